// Backward of one multi-head graph attention layer (GraphAttentionLayer / MultiHeadGATLayer, model/gat/graph_attention.py:40-160)
// in eval-dropout form (p = 0): what torch autograd computes for the reference when the graph branch is trained
// (scripts/train_end_to_end.py:219-226 puts the GAT parameters in the optimizer, :478 calls loss.backward()).
//
// Forward, per head (all heads in one pass; Wh = X W^T, s = Wh a_src, t = Wh a_tgt):
//     z_k = s_src(k) + t_tgt(k),  e_k = LeakyReLU(z_k),  x_k = exp(e_k - m),  m = max over ALL edges of the graph (:86)
//     D_j = sum_{k: tgt = j} x_k,  alpha_k = x_k / (D_j + 1e-10) (:96),  h'_j = sum_k alpha_k Wh_src(k),  out_j = ELU(h'_j) (:118)
// Backward (g* = gradient of the loss):
//     gh'_j  = gout_j * ELU'(h'_j)                       (gout / H for the head mean, :158)
//     ga_k   = gh'_tgt . Wh_src                          c_j = sum_k alpha_k ga_k
//     ge_k   = alpha_k (ga_k - c_j)                      gz_k = ge_k * (z_k > 0 ? 1 : slope)
//     gm     = -sum_j c_j * 1e-10 / (D_j + 1e-10)        (the global max is an input of every alpha; its gradient goes to the
//                                                         arg-max edge(s), evenly over ties -- torch.max()'s backward; it matters
//                                                         exactly where the 1e-10 bites: rows whose D_j is tiny)
//     gt_j   = sum_{k: tgt = j} gz_k                     gs_i = sum_{k: src = i} gz_k
//     gWh_i  = sum_{k: src = i} alpha_k gh'_tgt(k) + gs_i a_src + gt_i a_tgt
//     ga_src = sum_i gs_i Wh_i,  ga_tgt = sum_j gt_j Wh_j,  gW = gWh^T X,  gX = gWh W
// Kernels: one wavefront per target row (three passes over its in-edges; per-edge alpha and gz are parked in edge arrays),
// one block per (graph, head) for the max term, one wavefront per SOURCE row over the transposed CSR (no float atomics: every
// sum has a fixed order, so the gradients are bitwise reproducible), a two-level column reduction for ga, and the two GEMMs
// through the 1x1-convolution gradient launchers of the U-Net (bwd_blocks.hip).  HF = heads * Fh <= 256.
#include <algorithm>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "ctx.h"
#include "gat_common.h"

namespace mgu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gatb_edge_keys_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int N,
                                                             int* __restrict__ keys, int* __restrict__ vals, int* __restrict__ tgt) {
  // one wave per target row: key = source, value = edge position (target-CSR order), tgt[edge] = row
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= N) return;
  const int s0 = rowptr[j], s1 = rowptr[j + 1];
  for (int k = s0 + lane; k < s1; k += 64) keys[k] = col[k], vals[k] = k, tgt[k] = j;
}
__global__ __launch_bounds__(256) void gatb_rowptr_kernel(const int* __restrict__ sorted_keys, int64_t E, int N, int* __restrict__ rowptr) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > N) return;
  int64_t lo = 0, hi = E;   // first position with key >= j
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < j) lo = mid + 1; else hi = mid;
  }
  rowptr[j] = (int)lo;
}

// per (graph, head) max as a plain float table (decoded from the slotted accumulators of the forward's edge-max pass)
__global__ void gatb_gmax_decode_kernel(const gmax_t* __restrict__ gmax, int n, unsigned gen, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  gmax_t u = 0;
  for (int sl = 0; sl < GMAX_SLOTS; ++sl) {
    const gmax_t o = gmax[(size_t)e * GMAX_SLOTS + sl];
    u = o > u ? o : u;
  }
  out[e] = gmax_decode(u, gen);
}

// sum over the qh consecutive lanes of a head, valid in the head's FIRST lane (cin == 0): segmented doubling with shfl_down,
// any qh <= 64
__device__ __forceinline__ float head_sum(float v, int cin, int qh) {
  for (int o = 1; o < qh; o <<= 1) {
    const float u = __shfl_down(v, o);
    if (cin + o < qh) v += u;
  }
  return v;
}

// ---- per target row: alpha, gh', ga, c, gz, gt, the row's share of the max term -----------------------------------------------
__global__ __launch_bounds__(256) void gatb_target_kernel(const float* __restrict__ wh, const float* __restrict__ st,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ node_graph, const float* __restrict__ gm_f,
                                                          const float* __restrict__ dout, int N, int heads, int Fh, int concat, float slope,
                                                          float* __restrict__ ghp, float* __restrict__ alpha_e, float* __restrict__ gz_e,
                                                          float* __restrict__ gt, float* __restrict__ gmrow,
                                                          const float* __restrict__ edge_mask, const float* __restrict__ out_mask) {
  // edge_mask (E, heads), out_mask (N, F_out): the train-mode dropout masks of the attention coefficients and of the layer output
  // (graph_attention.py:97, :160; values 0 or 1 / (1 - p), NULL = none): h' = sum alpha_k m_k Wh_src, so d/d alpha_k carries m_k and
  // the softmax backward is unchanged (alpha itself is not masked)
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= N) return;   // wave-uniform
  const int HF = heads * Fh, nq = HF >> 2, qh = Fh >> 2, H2 = 2 * heads;
  const bool on = lane < nq;
  const int head = on ? lane / qh : 0, cin = on ? lane - head * qh : 0;
  const bool lead = on && cin == 0;                      // the lane that owns the (row, head) scalars
  const int s0 = rowptr[j], s1 = rowptr[j + 1];
  const int g = node_graph ? node_graph[j] : 0;
  const float tj = st[(size_t)j * H2 + heads + head], m = gm_f[g * heads + head];
  // pass 1: D, h'
  float D = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = s0; k < s1; ++k) {
    const int src = col[k];
    const float z = st[(size_t)src * H2 + head] + tj;
    const float x = on ? __expf((z > 0.f ? z : slope * z) - m) : 0.f;
    D += x;
    const float xm = edge_mask ? x * edge_mask[(size_t)k * heads + head] : x;
    acc += xm * *reinterpret_cast<const f32x4*>(wh + (size_t)src * HF + (on ? lane * 4 : 0));
  }
  const float inv = 1.f / (D + 1e-10f);
  f32x4 go = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    go = concat ? *reinterpret_cast<const f32x4*>(dout + (size_t)j * HF + lane * 4)
                : *reinterpret_cast<const f32x4*>(dout + (size_t)j * Fh + cin * 4) * (1.f / (float)heads);
    if (out_mask) go *= concat ? *reinterpret_cast<const f32x4*>(out_mask + (size_t)j * HF + lane * 4)
                               : *reinterpret_cast<const f32x4*>(out_mask + (size_t)j * Fh + cin * 4);
  }
  f32x4 gh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float hp = acc[q] * inv;
    gh[q] = go[q] * (hp > 0.f ? 1.f : __expf(hp));       // ELU'(x) = 1 (x > 0), exp(x) otherwise
  }
  if (on) *reinterpret_cast<f32x4*>(ghp + (size_t)j * HF + lane * 4) = gh;
  // pass 2: ga_k, alpha_k, c
  float c = 0.f;
  for (int k = s0; k < s1; ++k) {
    const int src = col[k];
    const f32x4 w = *reinterpret_cast<const f32x4*>(wh + (size_t)src * HF + (on ? lane * 4 : 0));
    float ga = head_sum(on ? gh[0] * w[0] + gh[1] * w[1] + gh[2] * w[2] + gh[3] * w[3] : 0.f, cin, qh);   // lead lane
    if (edge_mask) ga *= edge_mask[(size_t)k * heads + head];
    const float z = st[(size_t)src * H2 + head] + tj;
    const float al = __expf((z > 0.f ? z : slope * z) - m) * inv;
    c += al * ga;
    if (lead) alpha_e[(size_t)k * heads + head] = al, gz_e[(size_t)k * heads + head] = ga;
  }
  // pass 3: gz_k (own values back from the edge arrays: same lane wrote them), gt
  float gtv = 0.f;
  if (lead)
    for (int k = s0; k < s1; ++k) {
      const int src = col[k];
      const float z = st[(size_t)src * H2 + head] + tj;
      const float gz = alpha_e[(size_t)k * heads + head] * (gz_e[(size_t)k * heads + head] - c) * (z > 0.f ? 1.f : slope);
      gz_e[(size_t)k * heads + head] = gz;
      gtv += gz;
      // the source pass sums alpha_k m_k gh'_tgt: hand it the masked coefficient
      if (edge_mask) alpha_e[(size_t)k * heads + head] *= edge_mask[(size_t)k * heads + head];
    }
  if (lead) gt[(size_t)j * heads + head] = gtv, gmrow[(size_t)j * heads + head] = -c * 1e-10f * inv;
}

// ---- train-mode forward with explicit dropout masks (graph_attention.py:97, :160): one wavefront per target row -------------------
// h'_j = sum_k (alpha_k m_k) Wh_src(k), ELU, concat or head mean, times the output mask.  The same per-row walk as the backward's
// target pass (which recomputes exactly these values); the eval-mode schedules (gat_fused.hip, gat.hip) are not touched.
__global__ __launch_bounds__(256) void gatf_train_kernel(const float* __restrict__ wh, const float* __restrict__ st,
                                                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ node_graph, const float* __restrict__ gm_f, int N,
                                                         int heads, int Fh, int concat, float slope, const float* __restrict__ edge_mask,
                                                         const float* __restrict__ out_mask, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float stage[4][256];
  const int wv = threadIdx.x >> 6, j = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
  const bool row = j < N;                                  // wave-uniform; every wave reaches the barrier below
  const int jj = row ? j : N - 1;
  const int HF = heads * Fh, nq = HF >> 2, qh = Fh >> 2, H2 = 2 * heads;
  const bool on = lane < nq;
  const int head = on ? lane / qh : 0, cin = on ? lane - head * qh : 0;
  const int s0 = rowptr[jj], s1 = rowptr[jj + 1];
  const int g = node_graph ? node_graph[jj] : 0;
  const float tj = st[(size_t)jj * H2 + heads + head], m = gm_f[g * heads + head];
  float D = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = s0; k < s1; ++k) {
    const int src = col[k];
    const float z = st[(size_t)src * H2 + head] + tj;
    const float x = on ? __expf((z > 0.f ? z : slope * z) - m) : 0.f;   // exp(e - max(e)) (:86)
    D += x;
    const float xm = edge_mask ? x * edge_mask[(size_t)k * heads + head] : x;   // dropout of the coefficient (:97)
    acc += xm * *reinterpret_cast<const f32x4*>(wh + (size_t)src * HF + (on ? lane * 4 : 0));
  }
  const float inv = 1.f / (D + 1e-10f);                    // (:96)
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float hp = acc[q] * inv;
    v[q] = hp > 0.f ? hp : (__expf(hp) - 1.f);             // ELU (:118)
  }
  if (concat) {
    if (row && on) {
      if (out_mask) v *= *reinterpret_cast<const f32x4*>(out_mask + (size_t)j * HF + lane * 4);   // (:160)
      *reinterpret_cast<f32x4*>(out + (size_t)j * HF + lane * 4) = v;                               // cat over heads (:155)
    }
    return;
  }
  if (on) *reinterpret_cast<f32x4*>(&stage[wv][lane * 4]) = v;
  __syncthreads();
  if (row && lane < qh) {                                   // mean over heads (:158), fixed order
    f32x4 sum = *reinterpret_cast<const f32x4*>(&stage[wv][lane * 4]);
    for (int h = 1; h < heads; ++h) sum += *reinterpret_cast<const f32x4*>(&stage[wv][h * Fh + lane * 4]);
    sum *= 1.f / (float)heads;
    if (out_mask) sum *= *reinterpret_cast<const f32x4*>(out_mask + (size_t)j * Fh + lane * 4);      // (:160)
    *reinterpret_cast<f32x4*>(out + (size_t)j * Fh + lane * 4) = sum;
  }
}

// ---- dropout masks from a counter-based generator of the library's own (Philox-4x32-10): value i of stream `stream` under `seed` is a
// function of (seed, stream, i) alone -- reproducible, order-free, no state.  mask = u >= p ? 1 / (1 - p) : 0, nn.Dropout's rule.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], const unsigned k0, const unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (unsigned)p1, c[3] = (unsigned)p0, c[0] = n0, c[2] = n2;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned long long seed, unsigned long long stream, int64_t n, float p,
                                                           float* __restrict__ out) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;   // four values per counter
  if (i4 * 4 >= n) return;
  unsigned c[4] = {(unsigned)i4, (unsigned)((unsigned long long)i4 >> 32), (unsigned)stream, (unsigned)(stream >> 32)};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  const float keep = 1.f / (1.f - p);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i4 * 4 + e < n) out[i4 * 4 + e] = ((float)(c[e] >> 8) * (1.f / 16777216.f) >= p) ? keep : 0.f;
}

// ---- the max term: gm[g][h] = sum of the rows' shares (fixed order), added to the arg-max edge(s) of the graph -----------------
__global__ __launch_bounds__(256) void gatb_max_term_kernel(const float* __restrict__ st, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col, const int32_t* __restrict__ tgt_e,
                                                            const int32_t* __restrict__ graph_ptr, int N, int heads, float slope,
                                                            const float* __restrict__ gm_f, const float* __restrict__ gmrow,
                                                            float* __restrict__ gz_e, float* __restrict__ gt, int* __restrict__ err_word) {
  __shared__ double red[256];
  __shared__ int cnt;
  const int g = blockIdx.x / heads, h = blockIdx.x - g * heads, t = threadIdx.x, H2 = 2 * heads;
  const int n0 = graph_ptr ? graph_ptr[g] : 0, n1 = graph_ptr ? graph_ptr[g + 1] : N;
  double s = 0.0;
  for (int j = n0 + t; j < n1; j += 256) s += (double)gmrow[(size_t)j * heads + h];
  red[t] = s;
  if (t == 0) cnt = 0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float gm = (float)red[0], m = gm_f[g * heads + h];
  if (gm == 0.f) return;   // block-uniform
  // arg-max edges: compared in the ENCODED domain the forward's reduction works in (gat_enc_ordered: the maximum went through
  // encode -> atomicMax -> decode, a bijection on the bit pattern, so equal bits here == the edge that won there)
  const unsigned m_enc = gat_enc_ordered(m);
  const int e0 = rowptr[n0], e1 = rowptr[n1];
  int mine = 0;
  for (int k = e0 + t; k < e1; k += 256) {
    const float z = st[(size_t)col[k] * H2 + h] + st[(size_t)tgt_e[k] * H2 + heads + h];
    mine += (gat_enc_ordered(z > 0.f ? z : slope * z) == m_enc) ? 1 : 0;
  }
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (cnt == 0) {
    // no edge reproduces the forward's maximum (the score arithmetic here and in gat_edge_max_kernel must stay the same expression):
    // the term cannot be placed -- report it instead of dropping it silently
    if (t == 0 && err_word) atomicOr(err_word, 2);
    return;
  }
  const float share = gm / (float)cnt;   // evenly over ties (torch.max() backward)
  if (cnt == 1) {   // the usual case: one arg-max edge, one writer per element
    for (int k = e0 + t; k < e1; k += 256) {
      const float z = st[(size_t)col[k] * H2 + h] + st[(size_t)tgt_e[k] * H2 + heads + h];
      if (gat_enc_ordered(z > 0.f ? z : slope * z) == m_enc) {
        const float v = share * (z > 0.f ? 1.f : slope);
        gz_e[(size_t)k * heads + h] += v;
        gt[(size_t)tgt_e[k] * heads + h] += v;
      }
    }
  } else if (t == 0) {   // ties: one thread walks the graph's edges in order, so ties that share a target add in a fixed order
    for (int k = e0; k < e1; ++k) {
      const float z = st[(size_t)col[k] * H2 + h] + st[(size_t)tgt_e[k] * H2 + heads + h];
      if (gat_enc_ordered(z > 0.f ? z : slope * z) == m_enc) {
        const float v = share * (z > 0.f ? 1.f : slope);
        gz_e[(size_t)k * heads + h] += v;
        gt[(size_t)tgt_e[k] * heads + h] += v;
      }
    }
  }
}

// ---- per source row over the transposed CSR: gs, gWh ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gatb_source_kernel(const float* __restrict__ ghp, const float* __restrict__ alpha_e,
                                                          const float* __restrict__ gz_e, const float* __restrict__ gt,
                                                          const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ eid_s,
                                                          const int32_t* __restrict__ tgt_e, const float* __restrict__ a, int N, int heads,
                                                          int Fh, float* __restrict__ gwh, float* __restrict__ gs) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const int HF = heads * Fh, nq = HF >> 2, qh = Fh >> 2;
  const bool on = lane < nq;
  const int head = on ? lane / qh : 0, cin = on ? lane - head * qh : 0;
  float gsv = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = rowptr_s[i]; p < rowptr_s[i + 1]; ++p) {
    const int k = eid_s[p], j = tgt_e[k];
    gsv += gz_e[(size_t)k * heads + head];
    acc += alpha_e[(size_t)k * heads + head] * *reinterpret_cast<const f32x4*>(ghp + (size_t)j * HF + (on ? lane * 4 : 0));
  }
  if (!on) return;
  const float gtv = gt[(size_t)i * heads + head];
  const f32x4 as = *reinterpret_cast<const f32x4*>(a + (size_t)head * 2 * Fh + cin * 4);
  const f32x4 at = *reinterpret_cast<const f32x4*>(a + (size_t)head * 2 * Fh + Fh + cin * 4);
  *reinterpret_cast<f32x4*>(gwh + (size_t)i * HF + lane * 4) = acc + gsv * as + gtv * at;
  if (cin == 0) gs[(size_t)i * heads + head] = gsv;
}

// ---- ga: column sums of gs_i Wh_i and gt_i Wh_i (two levels, fixed order) ------------------------------------------------------
__global__ __launch_bounds__(256) void gatb_da_partial_kernel(const float* __restrict__ wh, const float* __restrict__ gs,
                                                              const float* __restrict__ gt, int N, int heads, int Fh, int rows_per_block,
                                                              float* __restrict__ part) {
  const int HF = heads * Fh, t = threadIdx.x;
  if (t >= HF) return;
  const int head = t / Fh;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  float ss = 0.f, tt = 0.f;
  for (int i = r0; i < r1; ++i) {
    const float w = wh[(size_t)i * HF + t];
    ss += gs[(size_t)i * heads + head] * w;
    tt += gt[(size_t)i * heads + head] * w;
  }
  part[((size_t)blockIdx.x * 2 + 0) * HF + t] = ss;
  part[((size_t)blockIdx.x * 2 + 1) * HF + t] = tt;
}
__global__ __launch_bounds__(256) void gatb_da_final_kernel(const float* __restrict__ part, int nb, int heads, int Fh, float* __restrict__ da) {
  const int HF = heads * Fh, t = threadIdx.x;
  if (t >= HF) return;
  double ss = 0.0, tt = 0.0;
  for (int b = 0; b < nb; ++b) ss += (double)part[((size_t)b * 2 + 0) * HF + t], tt += (double)part[((size_t)b * 2 + 1) * HF + t];
  const int head = t / Fh, f = t - head * Fh;
  da[(size_t)head * 2 * Fh + f] = (float)ss;        // a = [a_src | a_tgt] per head (graph_attention.py:31,61-64)
  da[(size_t)head * 2 * Fh + Fh + f] = (float)tt;
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

extern "C" {

int mgu_conv2d_dgrad_nhwc(mgu_ctx* c, const void* dz_dev, const void* w_oihw_dev, int B, int H, int W, int Cin, int Cout, int ksize,
                          void* din_dev, int ld_out, void* hip_stream);
int mgu_conv2d_wgrad_nhwc(mgu_ctx* c, const void* in_dev, int ld_in, const void* dz_dev, int B, int H, int W, int Cin, int Cout,
                          int ksize, void* dw_oihw_dev, void* hip_stream);

int mgu_csr_transpose_device(mgu_ctx* c, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E, int num_nodes,
                             int32_t* rowptr_src_dev, int32_t* eid_src_dev, int32_t* tgt_of_edge_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (E < 0 || num_nodes < 1 || E >= (1ll << 31) || !rowptr_dev || !rowptr_src_dev || (E > 0 && (!col_dev || !eid_src_dev || !tgt_of_edge_dev)))
    return fail(c, MGU_ERR_INVALID, "bad csr_transpose_device args (E < 2^31)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  if (E == 0) {
    HIPCHK(c, hipMemsetAsync(rowptr_src_dev, 0, ((size_t)num_nodes + 1) * sizeof(int32_t), s));
    return MGU_OK;
  }
  int bits = 1;
  while ((1ll << bits) < num_nodes) ++bits;
  size_t temp_bytes = 0;
  HIPCHK(c, rocprim::radix_sort_pairs(nullptr, temp_bytes, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)E, 0, bits, s));
  const size_t stride = ((size_t)E * 4 + 255) / 256 * 256;
  int rc = ensure(c, &c->gws, &c->gws_bytes, 3 * stride + temp_bytes + 256);
  if (rc) return rc;
  char* g = (char*)c->gws;
  int *keys = (int*)g, *vals = (int*)(g + stride), *skeys = (int*)(g + 2 * stride);
  hipLaunchKernelGGL(gatb_edge_keys_kernel, dim3((num_nodes + 3) / 4), dim3(256), 0, s, rowptr_dev, col_dev, num_nodes, keys, vals,
                     (int*)tgt_of_edge_dev);
  // stable LSD radix sort by source: a source's out-edges keep their target-CSR order -> every later sum has a fixed order
  HIPCHK(c, rocprim::radix_sort_pairs(g + 3 * stride, temp_bytes, keys, skeys, vals, (int*)eid_src_dev, (size_t)E, 0, bits, s));
  hipLaunchKernelGGL(gatb_rowptr_kernel, dim3((num_nodes + 1 + 255) / 256), dim3(256), 0, s, skeys, E, num_nodes, (int*)rowptr_src_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

static int gat_backward_impl(mgu_ctx* c, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                             const int32_t* rowptr_src_dev, const int32_t* eid_src_dev, const int32_t* tgt_of_edge_dev,
                             const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                             int concat, float alpha, const float* edge_mask_dev, const float* out_mask_dev, const void* dout_dev, void* dX_dev,
                             void* dW_dev, void* da_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  const int Fh = Fout_head, HF = heads * Fh;
  if (!X_dev || !rowptr_dev || !W_dev || !a_dev || !dout_dev || !dW_dev || !da_dev || N < 1 || Fin < 4 || (Fin & 3) || heads < 1 || Fh < 4 ||
      (Fh & 3) || E < 0 || (E > 0 && (!col_dev || !rowptr_src_dev || !eid_src_dev || !tgt_of_edge_dev)))
    return fail(c, MGU_ERR_INVALID, "bad gat_layer_backward args (Fin, Fout_head multiples of 4)");
  if (HF > 256) return fail(c, MGU_ERR_INVALID, "gat_layer_backward supports heads * Fout_head <= 256 (got %d x %d)", heads, Fh);
  if ((int64_t)N * HF >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "N * heads * Fout_head must be < 2^31");
  if (num_graphs < 1 || !graph_ptr_dev) num_graphs = 1, graph_ptr_dev = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  // ---- scratch of its own (the GEMM launchers below use the shared building-block scratch) ----
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const int nb = std::max(1, std::min(256, (N + 255) / 256));
  const int rpb = (N + nb - 1) / nb;
  const size_t o_wh = take((size_t)N * HF * 4), o_st = take((size_t)N * 2 * heads * 4), o_ng = take((size_t)N * 4),
               o_gm = take((size_t)num_graphs * heads * 4), o_ghp = take((size_t)N * HF * 4), o_al = take((size_t)std::max<int64_t>(E, 1) * heads * 4),
               o_gz = take((size_t)std::max<int64_t>(E, 1) * heads * 4), o_gt = take((size_t)N * heads * 4), o_gs = take((size_t)N * heads * 4),
               o_gmr = take((size_t)N * heads * 4), o_gwh = take((size_t)N * HF * 4), o_part = take((size_t)nb * 2 * HF * 4);
  int rc = ensure(c, &c->gbws, &c->gbws_bytes, off);
  if (rc) return rc;
  char* g = (char*)c->gbws;
  float *wh = (float*)(g + o_wh), *st = (float*)(g + o_st), *gm_f = (float*)(g + o_gm), *ghp = (float*)(g + o_ghp), *al = (float*)(g + o_al),
        *gz = (float*)(g + o_gz), *gt = (float*)(g + o_gt), *gs = (float*)(g + o_gs), *gmr = (float*)(g + o_gmr), *gwh = (float*)(g + o_gwh),
        *part = (float*)(g + o_part);
  int32_t* node_graph = num_graphs > 1 ? (int32_t*)(g + o_ng) : nullptr;
  // ---- recompute Wh, s, t and the per-graph max exactly as the forward's gather schedule does ----
  if ((rc = gat_linear_st(c, (const float*)X_dev, N, Fin, (const float*)W_dev, (const float*)a_dev, heads, Fh, wh, st, s))) return rc;
  unsigned long long* gmax;
  unsigned gen;
  if ((rc = gmax_buffer(c, num_graphs * heads, &gmax, &gen))) return rc;
  if (node_graph) HIPCHK(c, launch_gat_node_graph(graph_ptr_dev, num_graphs, 0, N, node_graph, s));
  HIPCHK(c, launch_gat_edge_max(st, rowptr_dev, col_dev, node_graph, N, heads, alpha, gmax, c->gmax_cap, gen, s));
  hipLaunchKernelGGL(gatb_gmax_decode_kernel, dim3((num_graphs * heads + 63) / 64), dim3(64), 0, s, gmax, num_graphs * heads, gen, gm_f);
  // ---- attention backward ----
  hipLaunchKernelGGL(gatb_target_kernel, dim3((N + 3) / 4), dim3(256), 0, s, wh, st, rowptr_dev, col_dev, node_graph, gm_f,
                     (const float*)dout_dev, N, heads, Fh, concat, alpha, ghp, al, gz, gt, gmr, edge_mask_dev, out_mask_dev);
  if (E > 0) {
    int* err_dev = nullptr;
    if ((rc = err_word_dev(c, &err_dev))) return rc;
    hipLaunchKernelGGL(gatb_max_term_kernel, dim3(num_graphs * heads), dim3(256), 0, s, st, rowptr_dev, col_dev, tgt_of_edge_dev, graph_ptr_dev, N,
                       heads, alpha, gm_f, gmr, gz, gt, err_dev);
  }
  if (E > 0) {
    hipLaunchKernelGGL(gatb_source_kernel, dim3((N + 3) / 4), dim3(256), 0, s, ghp, al, gz, gt, rowptr_src_dev, eid_src_dev, tgt_of_edge_dev,
                       (const float*)a_dev, N, heads, Fh, gwh, gs);
  } else {   // no edges: nothing reaches Wh (every output row is ELU(0) = 0)
    HIPCHK(c, hipMemsetAsync(gwh, 0, (size_t)N * HF * 4, s));
    HIPCHK(c, hipMemsetAsync(gs, 0, (size_t)N * heads * 4, s));
  }
  hipLaunchKernelGGL(gatb_da_partial_kernel, dim3(nb), dim3(256), 0, s, wh, gs, gt, N, heads, Fh, rpb, part);
  hipLaunchKernelGGL(gatb_da_final_kernel, dim3(1), dim3(256), 0, s, part, nb, heads, Fh, (float*)da_dev);
  HIPCHK(c, hipGetLastError());
  // ---- the linear layer: gW = gWh^T X, gX = gWh W (a 1x1 convolution over an N x 1 image) ----
  if ((rc = mgu_conv2d_wgrad_nhwc(c, X_dev, Fin, gwh, 1, 1, N, Fin, HF, 1, dW_dev, hip_stream))) return rc;
  if (dX_dev && (rc = mgu_conv2d_dgrad_nhwc(c, gwh, W_dev, 1, 1, N, Fin, HF, 1, dX_dev, Fin, hip_stream))) return rc;
  return MGU_OK;
}

int mgu_gat_layer_backward(mgu_ctx* c, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                           const int32_t* rowptr_src_dev, const int32_t* eid_src_dev, const int32_t* tgt_of_edge_dev,
                           const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                           int concat, float alpha, const void* dout_dev, void* dX_dev, void* dW_dev, void* da_dev, void* hip_stream) {
  return gat_backward_impl(c, X_dev, N, Fin, rowptr_dev, col_dev, E, rowptr_src_dev, eid_src_dev, tgt_of_edge_dev, graph_ptr_dev, num_graphs, W_dev,
                           a_dev, heads, Fout_head, concat, alpha, nullptr, nullptr, dout_dev, dX_dev, dW_dev, da_dev, hip_stream);
}

int mgu_gat_layer_backward_train(mgu_ctx* c, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                                 const int32_t* rowptr_src_dev, const int32_t* eid_src_dev, const int32_t* tgt_of_edge_dev,
                                 const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                                 int concat, float alpha, const void* edge_mask_dev, const void* out_mask_dev, const void* dout_dev,
                                 void* dX_dev, void* dW_dev, void* da_dev, void* hip_stream) {
  return gat_backward_impl(c, X_dev, N, Fin, rowptr_dev, col_dev, E, rowptr_src_dev, eid_src_dev, tgt_of_edge_dev, graph_ptr_dev, num_graphs, W_dev,
                           a_dev, heads, Fout_head, concat, alpha, (const float*)edge_mask_dev, (const float*)out_mask_dev, dout_dev, dX_dev, dW_dev,
                           da_dev, hip_stream);
}

int mgu_gat_layer_forward_train(mgu_ctx* c, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                                const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                                int concat, float alpha, const void* edge_mask_dev, const void* out_mask_dev, void* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  const int Fh = Fout_head, HF = heads * Fh;
  if (!X_dev || !rowptr_dev || !W_dev || !a_dev || !out_dev || N < 1 || Fin < 4 || (Fin & 3) || heads < 1 || Fh < 4 || (Fh & 3) || E < 0 ||
      (E > 0 && !col_dev))
    return fail(c, MGU_ERR_INVALID, "bad gat_layer_forward_train args (Fin, Fout_head multiples of 4)");
  if (HF > 256) return fail(c, MGU_ERR_INVALID, "gat_layer_forward_train supports heads * Fout_head <= 256 (got %d x %d)", heads, Fh);
  if ((int64_t)N * HF >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "N * heads * Fout_head must be < 2^31");
  if (num_graphs < 1 || !graph_ptr_dev) num_graphs = 1, graph_ptr_dev = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t o_wh = take((size_t)N * HF * 4), o_st = take((size_t)N * 2 * heads * 4), o_ng = take((size_t)N * 4),
               o_gm = take((size_t)num_graphs * heads * 4);
  int rc = ensure(c, &c->gbws, &c->gbws_bytes, off);
  if (rc) return rc;
  char* g = (char*)c->gbws;
  float *wh = (float*)(g + o_wh), *st = (float*)(g + o_st), *gm_f = (float*)(g + o_gm);
  int32_t* node_graph = num_graphs > 1 ? (int32_t*)(g + o_ng) : nullptr;
  // Wh, s, t from one GEMM and the per-graph max, exactly as mgu_gat_layer_backward recomputes them
  if ((rc = gat_linear_st(c, (const float*)X_dev, N, Fin, (const float*)W_dev, (const float*)a_dev, heads, Fh, wh, st, s))) return rc;
  unsigned long long* gmax;
  unsigned gen;
  if ((rc = gmax_buffer(c, num_graphs * heads, &gmax, &gen))) return rc;
  if (node_graph) HIPCHK(c, launch_gat_node_graph(graph_ptr_dev, num_graphs, 0, N, node_graph, s));
  if (E > 0) HIPCHK(c, launch_gat_edge_max(st, rowptr_dev, col_dev, node_graph, N, heads, alpha, gmax, c->gmax_cap, gen, s));
  hipLaunchKernelGGL(gatb_gmax_decode_kernel, dim3((num_graphs * heads + 63) / 64), dim3(64), 0, s, gmax, num_graphs * heads, gen, gm_f);
  hipLaunchKernelGGL(gatf_train_kernel, dim3((N + 3) / 4), dim3(256), 0, s, wh, st, rowptr_dev, col_dev, node_graph, gm_f, N, heads, Fh, concat,
                     alpha, (const float*)edge_mask_dev, (const float*)out_mask_dev, (float*)out_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_dropout_mask(mgu_ctx* c, unsigned long long seed, unsigned long long stream, int64_t n, float p, void* mask_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (n < 0 || !(p >= 0.f && p < 1.f) || (n > 0 && !mask_dev)) return fail(c, MGU_ERR_INVALID, "bad dropout_mask args (0 <= p < 1)");
  if (n == 0) return MGU_OK;
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)hip_stream, seed, stream, n, p, (float*)mask_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

}  // extern "C"
