// Graph attention message passing for gfx950: replaces GraphAttentionLayer.forward
// (model/gat/graph_attention.py:40-118) and the per-head Python loop of MultiHeadGATLayer (:150-160).
//
// The reference gathers 2*E rows of Wh to build e = LeakyReLU(a . [Wh_src || Wh_tgt]) (:57-65).  That
// dot product splits into per-NODE scalars s_i = a[:F'] . Wh_i and t_j = a[F':] . Wh_j, so an edge only
// needs two scalars for its logit and ONE row gather (Wh_src) for the aggregate.  All heads are done in
// one pass: a row of Wh is heads*F' floats (1 KiB for 4x64), i.e. exactly one 16-byte-per-lane
// coalesced wavefront load.  The softmax is the reference's literal form: exp(e - GLOBAL max over the
// graph's edges) (:86), divided by (segment sum + 1e-10) (:96) -- not a per-target stabilised softmax.
//
//   gat_st        : s,t per node and head                         (reads Wh once)
//   gat_edge_max  : per (graph, head) max_e                        (CSR + s,t only; order-encoded atomicMax)
//   gat_aggregate : CSR-by-target neighbour gather -> weighted sum -> /(D+1e-10) -> ELU -> concat | head-mean
//                   one wavefront per target row; neighbour ids are fetched 64 at a time (one per
//                   lane) and broadcast with v_readlane, the head-mean is staged through LDS.
#include "common.h"

namespace mgu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned enc_ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ int graph_of(const int32_t* __restrict__ gp, int G, int node) {
  if (!gp || G <= 1) return 0;
  int lo = 0, hi = G;  // gp[lo] <= node < gp[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gp[mid] <= node) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- s_i[h] = a[h][:Fh] . Wh_i[h], t_i[h] = a[h][Fh:] . Wh_i[h]  (graph_attention.py:61-64, split) ----
constexpr int GAT_MAX_HF = 1024;

__global__ __launch_bounds__(256) void gat_st_kernel(const float* __restrict__ Wh, const float* __restrict__ a,
                                                     float* __restrict__ st, int N, int heads, int Fh) {
  __shared__ float part[4][2][GAT_MAX_HF / 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int node = blockIdx.x * 4 + wave;
  const int HF = heads * Fh, nq = HF >> 2, qh = Fh >> 2;
  if (node < N) {
    for (int c = lane; c < nq; c += 64) {
      const int h = c / qh, f = (c - h * qh) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wh + (size_t)node * HF + c * 4);
      const f32x4 as = *reinterpret_cast<const f32x4*>(a + h * 2 * Fh + f);
      const f32x4 at = *reinterpret_cast<const f32x4*>(a + h * 2 * Fh + Fh + f);
      part[wave][0][c] = v[0] * as[0] + v[1] * as[1] + v[2] * as[2] + v[3] * as[3];
      part[wave][1][c] = v[0] * at[0] + v[1] * at[1] + v[2] * at[2] + v[3] * at[3];
    }
  }
  __syncthreads();
  if (node < N && lane < 2 * heads) {
    const int which = lane / heads, h = lane - which * heads;
    float sum = 0.f;
    for (int c = h * qh; c < (h + 1) * qh; ++c) sum += part[wave][which][c];
    st[(size_t)which * N * heads + (size_t)node * heads + h] = sum;
  }
}

hipError_t launch_gat_st(const float* Wh, const float* a, float* st, int N, int heads, int Fh, hipStream_t s) {
  if (heads * Fh > GAT_MAX_HF || (Fh & 3) || heads > 32) return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  hipLaunchKernelGGL(gat_st_kernel, dim3((N + 3) / 4), dim3(256), 0, s, Wh, a, st, N, heads, Fh);
  return hipGetLastError();
}

// ---- per (graph, head): max over edges of LeakyReLU(s_src + t_tgt)  (torch.max(e), :86) -----------
__global__ __launch_bounds__(256) void gat_edge_max_kernel(const float* __restrict__ st, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ gp, int G, int N, int heads,
                                                           float alpha, unsigned* __restrict__ gmax) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave_first = j - lane;
  if (wave_first >= N) return;  // wave-uniform
  const bool valid = j < N;
  const int g = graph_of(gp, G, valid ? j : wave_first);
  const int g0 = __builtin_amdgcn_readfirstlane(g);
  const bool uniform = __all(g == g0);
  const float* s = st;
  const float* t = st + (size_t)N * heads;
  int start = 0, end = 0;
  if (valid) {
    start = rowptr[j];
    end = rowptr[j + 1];
  }
  for (int h = 0; h < heads; ++h) {
    float m = -INFINITY;
    for (int k = start; k < end; ++k) m = fmaxf(m, s[(size_t)col[k] * heads + h]);
    float e = -INFINITY;
    if (end > start) {
      e = m + t[(size_t)j * heads + h];
      e = e > 0.f ? e : alpha * e;  // LeakyReLU is monotone: max commutes with it
    }
    if (uniform) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) e = fmaxf(e, __shfl_xor(e, off));
      if (lane == 0 && e > -INFINITY) atomicMax(&gmax[g0 * heads + h], enc_ordered(e));
    } else if (e > -INFINITY) {
      atomicMax(&gmax[g * heads + h], enc_ordered(e));
    }
  }
}

hipError_t launch_gat_edge_max(const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* graph_ptr,
                               int num_graphs, int N, int heads, float alpha, unsigned* gmax_enc, hipStream_t s) {
  if (N == 0) return hipSuccess;
  hipLaunchKernelGGL(gat_edge_max_kernel, dim3((N + 255) / 256), dim3(256), 0, s, st, rowptr, col, graph_ptr,
                     num_graphs, N, heads, alpha, gmax_enc);
  return hipGetLastError();
}

// ---- neighbour gather + attention-weighted aggregate + normalise + ELU + concat/mean --------------
template <int NCH>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(const float* __restrict__ Wh, const float* __restrict__ st,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ gp, int G,
                                                            const unsigned* __restrict__ gmax, int N, int heads, int Fh,
                                                            int concat, float alpha, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float stage[4][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= N) return;  // wave-uniform; no block-level barrier below
  const int HF = heads * Fh, nq = HF >> 2, qh = Fh >> 2;
  const int g = graph_of(gp, G, j);
  const float* s = st;
  const float* t = st + (size_t)N * heads;
  const int start = rowptr[j], end = rowptr[j + 1];

  int head[NCH];
  float tj[NCH], gm[NCH], D[NCH];
  f32x4 acc[NCH];
  bool on[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    on[i] = c < nq;
    head[i] = on[i] ? c / qh : 0;
    tj[i] = t[(size_t)j * heads + head[i]];
    gm[i] = dec_ordered(gmax[g * heads + head[i]]);
    D[i] = 0.f;
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int base = start; base < end; base += 64) {
    const int mine = (base + lane < end) ? col[base + lane] : 0;
    const int cnt = min(64, end - base);
    for (int e = 0; e < cnt; ++e) {
      const int src = __builtin_amdgcn_readlane(mine, e);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (on[i]) {
          float ev = s[(size_t)src * heads + head[i]] + tj[i];
          ev = ev > 0.f ? ev : alpha * ev;                       // LeakyReLU (:65)
          const float x = expf(ev - gm[i]);                      // exp(e - max(e)) (:86)
          const f32x4 v = *reinterpret_cast<const f32x4*>(Wh + (size_t)src * HF + (lane + 64 * i) * 4);
          D[i] += x;                                             // scatter_add of exp_e (:90-91)
          acc[i] += x * v;                                       // scatter_add of alpha*Wh_src (:104-112)
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (on[i]) {
      const float inv = 1.f / (D[i] + 1e-10f);                   // (:96)
      f32x4 o = acc[i] * inv;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : expm1f(o[q]);  // ELU (:118)
      if (concat)
        *reinterpret_cast<f32x4*>(out + (size_t)j * HF + (lane + 64 * i) * 4) = o;   // torch.cat (:155)
      else
        *reinterpret_cast<f32x4*>(&stage[wave][(lane + 64 * i) * 4]) = o;
    }
  }
  if (!concat) {
    // head mean (:158): lanes 0..Fh/4-1 each own one float4 of the output row.  Only this wave touches
    // stage[wave], and a wavefront's LDS ops complete in order, so no barrier is required.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int c = lane; c < qh; c += 64) {
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int h = 0; h < heads; ++h) sum += *reinterpret_cast<const f32x4*>(&stage[wave][h * Fh + c * 4]);
      *reinterpret_cast<f32x4*>(out + (size_t)j * Fh + c * 4) = sum / (float)heads;
    }
  }
}

hipError_t launch_gat_aggregate(const float* Wh, const float* st, const int32_t* rowptr, const int32_t* col,
                                const int32_t* graph_ptr, int num_graphs, const unsigned* gmax_enc, int N, int heads,
                                int Fh, int concat, float alpha, float* out, hipStream_t s) {
  const int HF = heads * Fh;
  if (HF > GAT_MAX_HF || (Fh & 3)) return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  dim3 grid((N + 3) / 4), block(256);
#define MGU_AGG(NCH)                                                                                             \
  hipLaunchKernelGGL(gat_aggregate_kernel<NCH>, grid, block, 0, s, Wh, st, rowptr, col, graph_ptr, num_graphs,   \
                     gmax_enc, N, heads, Fh, concat, alpha, out)
  if (HF <= 256) MGU_AGG(1);
  else if (HF <= 512) MGU_AGG(2);
  else MGU_AGG(4);
#undef MGU_AGG
  return hipGetLastError();
}

}  // namespace mgu
