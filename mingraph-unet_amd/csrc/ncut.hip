// MinCut stage of the patch-graph branch: MinCutRefinement.normalized_cut_loss / .forward
// (model/graph_partition/mincut_refinement.py:30-52, 55-160, 188-205) on the same CSR gather primitive as the GAT.
//
//   w_e      = exp(-|f_src - f_tgt|^2 / 2)                                   (:43-51, sigma = 1)
//   deg_i    = sum over edges LEAVING i of w_e                               (:93-96: scatter_add over the SOURCE index)
//   assoc_k  = sum_i P_ik deg_i                                              (:104)
//   cut_k    = sum_e w_e P_src,k (1 - P_tgt,k)                               (:116-117, :150)
//   loss     = sum_k cut_k / assoc_k  over the segments with assoc_k > 1e-8  (:152-153)
//
// The reference recomputes the degree vector inside its K loop and gathers both end points of every edge three times;
// here one wavefront owns a SOURCE node (CSR by source): its feature row stays in registers, each out-edge costs one
// gathered row of the target (256 bytes for 64 features, a coalesced wave load) and a wave reduction, and the node's
// contributions to all K cuts and associations are formed at once -- P_ik factors out of both sums.  Per-node
// partial sums meet in doubles (one atomic per workgroup and quantity), so the result does not depend on the order.
#include "ctx.h"

namespace mgu {

constexpr int NCUT_MAX_K = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// softmax over the K segment logits of a node (:190) and the hard label train_end_to_end.py:356 takes from it
__global__ void ncut_softmax_kernel(const float* __restrict__ logits, int N, int K, float* __restrict__ soft,
                                    int32_t* __restrict__ hard) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* l = logits + (size_t)i * K;
  float m = l[0];
  int am = 0;
  for (int k = 1; k < K; ++k)
    if (l[k] > m) m = l[k], am = k;   // first maximum, like torch.argmax
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(l[k] - m);
  const float inv = 1.f / s;
  for (int k = 0; k < K; ++k) soft[(size_t)i * K + k] = expf(l[k] - m) * inv;
  if (hard) hard[i] = am;
}

// acc[slot][0..K) += cut_k, acc[slot][K..2K) += assoc_k, slot = workgroup % NCUT_SLOTS: thousands of workgroups adding
// into the same 2K doubles serialise in the L2 atomic unit (measured: 204 us for 65 536 nodes); 64 slots do not.
constexpr int NCUT_SLOTS = 64;

__global__ __launch_bounds__(256) void ncut_node_kernel(const float* __restrict__ F, int N, int D, const float* __restrict__ P, int K,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        double* __restrict__ acc) {
  __shared__ float part[4][2 * NCUT_MAX_K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  float cutp[NCUT_MAX_K];
#pragma unroll
  for (int k = 0; k < NCUT_MAX_K; ++k) cutp[k] = 0.f;
  float deg = 0.f;
  if (i < N) {   // wave-uniform
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const float* fi = F + (size_t)i * D;
    // four out-edges at a time (a patch-graph node has at most four): the target rows are gathered before any of them
    // is reduced, so the four row loads and the four wave reductions overlap; slots past the end re-read the last edge
    // and get weight 0
    for (int e = e0; e < e1; e += 4) {
      int t[4];
      float d2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = col[min(e + u, e1 - 1)];
      for (int c = lane; c < D; c += 64) {
        const float x = fi[c];
        float y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = F[(size_t)t[u] * D + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) d2[u] += (x - y[u]) * (x - y[u]);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) d2[u] += __shfl_xor(d2[u], off);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = (e + u < e1) ? expf(-0.5f * d2[u]) : 0.f;
        deg += w;
#pragma unroll
        for (int k = 0; k < NCUT_MAX_K; ++k)
          if (k < K) cutp[k] += w * (1.f - P[(size_t)t[u] * K + k]);
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NCUT_MAX_K; ++k) {
      const float p = (i < N && k < K) ? P[(size_t)i * K + k] : 0.f;
      part[wave][k] = p * cutp[k];
      part[wave][NCUT_MAX_K + k] = p * deg;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * K) {
    const int which = threadIdx.x / K, k = threadIdx.x - which * K;
    const double s = (double)part[0][which * NCUT_MAX_K + k] + (double)part[1][which * NCUT_MAX_K + k] +
                     (double)part[2][which * NCUT_MAX_K + k] + (double)part[3][which * NCUT_MAX_K + k];
    atomicAdd(acc + (size_t)(blockIdx.x % NCUT_SLOTS) * 2 * K + which * K + k, s);
  }
}

// D % 4 == 0 (every caller on the path: 64 features): SIXTEEN lanes own a node, each holding four features, so a wavefront
// walks four nodes at once and a 256-byte feature row is one 16-byte load per lane.  The kernel is a chain of dependent
// gathers (rowptr -> col -> rows); four nodes per wave quadruple the loads in flight per wave.
typedef float f32x4n __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ncut_node16_kernel(const float* __restrict__ F, int N, int D, const float* __restrict__ P, int K,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          double* __restrict__ acc) {
  __shared__ float part[16][2 * NCUT_MAX_K + 1];
  const int gl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + grp;
  const bool live = i < N;
  float cutp[NCUT_MAX_K];
#pragma unroll
  for (int k = 0; k < NCUT_MAX_K; ++k) cutp[k] = 0.f;
  float deg = 0.f;
  const int e0 = live ? rowptr[i] : 0, e1 = live ? rowptr[i + 1] : 0;
  const float* fi = F + (size_t)(live ? i : 0) * D;
  int emax = e1 - e0;   // the longest edge list of the wave decides the trip count (shuffles need all lanes)
#pragma unroll
  for (int off = 32; off >= 16; off >>= 1) emax = max(emax, __shfl_xor(emax, off));
  for (int eb = 0; eb < emax; eb += 4) {
    int t[4];
    float d2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = (e0 + eb + u < e1) ? col[e0 + eb + u] : (live ? i : 0);
    for (int c = gl * 4; c < D; c += 64) {
      const f32x4n x = *reinterpret_cast<const f32x4n*>(fi + c);
      f32x4n y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) y[u] = *reinterpret_cast<const f32x4n*>(F + (size_t)t[u] * D + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4n d = x - y[u];
        d2[u] += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) d2[u] += __shfl_xor(d2[u], off);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = (e0 + eb + u < e1) ? expf(-0.5f * d2[u]) : 0.f;
      deg += w;
#pragma unroll
      for (int k = 0; k < NCUT_MAX_K; ++k)
        if (k < K) cutp[k] += w * (1.f - P[(size_t)t[u] * K + k]);
    }
  }
  if (gl == 0) {
#pragma unroll
    for (int k = 0; k < NCUT_MAX_K; ++k) {
      const float p = (live && k < K) ? P[(size_t)i * K + k] : 0.f;
      part[grp][k] = p * cutp[k];
      part[grp][NCUT_MAX_K + k] = p * deg;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * K) {
    const int which = threadIdx.x / K, k = threadIdx.x - which * K;
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += (double)part[g][which * NCUT_MAX_K + k];
    atomicAdd(acc + (size_t)(blockIdx.x % NCUT_SLOTS) * 2 * K + which * K + k, s);
  }
}

__global__ void ncut_finalize_kernel(const double* __restrict__ acc, int K, float* __restrict__ loss) {
  __shared__ double sums[2 * NCUT_MAX_K];
  if (threadIdx.x < 2 * K) {
    double s = 0.0;
    for (int sl = 0; sl < NCUT_SLOTS; ++sl) s += acc[(size_t)sl * 2 * K + threadIdx.x];
    sums[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x) return;
  float total = 0.f;
  for (int k = 0; k < K; ++k) {
    const float cut = (float)sums[k], assoc = (float)sums[K + k];
    if (assoc > 1e-8f) total += cut / assoc;   // :152-153
  }
  *loss = total;
}

// ---- backward of the loss (L_partition.backward() of scripts/train_end_to_end.py:348-356, 472-479) ------------------------------
// With c_k = cut_k, a_k = assoc_k, L = sum_k c_k / a_k over the kept segments and g the upstream gradient:
//   alpha_k = g / a_k,  beta_k = -g c_k / a_k^2                                     (0 for a skipped segment: it has no term)
//   dL/dP_ik = sum_{e: src = i} w_e (alpha_k (1 - P_tgt,k) + beta_k)  -  sum_{e: tgt = i} w_e alpha_k P_src,k
//   dL/dw_e  = sum_k P_src,k (alpha_k (1 - P_tgt,k) + beta_k) =: q_e,   dw_e/df_src = -w_e (f_src - f_tgt) = -dw_e/df_tgt
//   dL/df_i  = -sum_{e: src = i} q_e w_e (f_i - f_tgt)  +  sum_{e: tgt = i} q_e w_e (f_src - f_i)
// (the reference differentiates through the edge weights: node_features reach compute_edge_weights_for_ncut undetached, :79).
// One wavefront owns node i and GATHERS over both of its edge lists (CSR by source and CSR by target), so no gradient is
// scattered and the result does not depend on a summation order; the weights are recomputed (a row distance per edge).
constexpr int NCUT_BWD_DREGS = 16;   // feature registers per lane: D <= 1024

__global__ void ncut_coef_kernel(const double* __restrict__ acc, int K, const float* __restrict__ gloss, float* __restrict__ coef) {
  const int k = threadIdx.x;
  if (k >= K) return;
  double c = 0.0, a = 0.0;
  for (int sl = 0; sl < NCUT_SLOTS; ++sl) c += acc[(size_t)sl * 2 * K + k], a += acc[(size_t)sl * 2 * K + K + k];
  const float cut = (float)c, assoc = (float)a, g = gloss ? *gloss : 1.f;
  const bool kept = assoc > 1e-8f;   // :152
  coef[k] = kept ? g / assoc : 0.f;
  coef[NCUT_MAX_K + k] = kept ? -g * cut / (assoc * assoc) : 0.f;
}

__global__ __launch_bounds__(256) void ncut_bwd_node_kernel(const float* __restrict__ F, int N, int D, const float* __restrict__ P, int K,
                                                            const int32_t* __restrict__ rp_src, const int32_t* __restrict__ col_tgt,
                                                            const int32_t* __restrict__ rp_tgt, const int32_t* __restrict__ col_src,
                                                            const float* __restrict__ coef, const float* __restrict__ gsoft,
                                                            int is_logits, float* __restrict__ dA, float* __restrict__ dF) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;   // wave-uniform, no barrier below
  float Pi[NCUT_MAX_K], al[NCUT_MAX_K], be[NCUT_MAX_K], dP[NCUT_MAX_K];
#pragma unroll
  for (int k = 0; k < NCUT_MAX_K; ++k) {
    Pi[k] = k < K ? P[(size_t)i * K + k] : 0.f;
    al[k] = k < K ? coef[k] : 0.f;
    be[k] = k < K ? coef[NCUT_MAX_K + k] : 0.f;
    dP[k] = 0.f;
  }
  float fi[NCUT_BWD_DREGS], df[NCUT_BWD_DREGS];
#pragma unroll
  for (int r = 0; r < NCUT_BWD_DREGS; ++r) {
    const int c = lane + 64 * r;
    fi[r] = c < D ? F[(size_t)i * D + c] : 0.f;
    df[r] = 0.f;
  }
  // both[0] = edges leaving i (other end = target), both[1] = edges entering i (other end = source)
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    const int32_t* rp = dir ? rp_tgt : rp_src;
    const int32_t* cl = dir ? col_src : col_tgt;
    const int e1 = rp[i + 1];
    for (int e = rp[i]; e < e1; ++e) {
      const int o = cl[e];
      float fo[NCUT_BWD_DREGS];
      float d2 = 0.f;
#pragma unroll
      for (int r = 0; r < NCUT_BWD_DREGS; ++r) {
        const int c = lane + 64 * r;
        fo[r] = c < D ? F[(size_t)o * D + c] : 0.f;
        d2 += (fi[r] - fo[r]) * (fi[r] - fo[r]);
      }
      d2 = wave_sum(d2);
      const float w = expf(-0.5f * d2);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NCUT_MAX_K; ++k) {
        if (k < K) {
          const float Po = P[(size_t)o * K + k];
          if (dir == 0) {   // i is the source
            const float t = al[k] * (1.f - Po) + be[k];
            q += Pi[k] * t;
            dP[k] += w * t;
          } else {          // i is the target
            q += Po * (al[k] * (1.f - Pi[k]) + be[k]);
            dP[k] -= w * al[k] * Po;
          }
        }
      }
      // dir 0: -q w (f_i - f_o);  dir 1: +q w (f_o - f_i): the same expression
      const float qw = q * w;
#pragma unroll
      for (int r = 0; r < NCUT_BWD_DREGS; ++r) df[r] += qw * (fo[r] - fi[r]);
    }
  }
  if (dF) {
#pragma unroll
    for (int r = 0; r < NCUT_BWD_DREGS; ++r) {
      const int c = lane + 64 * r;
      if (c < D) dF[(size_t)i * D + c] = df[r];
    }
  }
  if (lane == 0) {
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NCUT_MAX_K; ++k) {
      if (k < K && gsoft) dP[k] += gsoft[(size_t)i * K + k];   // gradient arriving through the returned soft assignments
      dot += Pi[k] * dP[k];
    }
#pragma unroll
    for (int k = 0; k < NCUT_MAX_K; ++k)
      if (k < K) dA[(size_t)i * K + k] = is_logits ? Pi[k] * (dP[k] - dot) : dP[k];   // softmax backward (:191)
  }
}

// dz = dy where y > 0 (the ReLU between the two Linear layers of the MLP segment predictor, train_end_to_end.py:59-63)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// one wavefront per edge of the COO list
__global__ __launch_bounds__(256) void ncut_edge_weight_kernel(const float* __restrict__ F, int D, const int64_t* __restrict__ src,
                                                               const int64_t* __restrict__ tgt, int64_t E, float* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const float* fs = F + (size_t)src[e] * D;
  const float* ft = F + (size_t)tgt[e] * D;
  float d2 = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = fs[c] - ft[c];
    d2 += d * d;
  }
  d2 = wave_sum(d2);
  if (lane == 0) w[e] = expf(-0.5f * d2);
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

extern "C" {

int mgu_ncut_edge_weights(mgu_ctx* c, const float* feats_dev, int N, int D, const int64_t* edge_index_dev, int64_t E, float* w_dev,
                          void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (N < 0 || D <= 0 || E < 0) return fail(c, MGU_ERR_INVALID, "mgu_ncut_edge_weights: bad sizes N=%d D=%d E=%lld", N, D, (long long)E);
  if (E == 0) return MGU_OK;
  if (!feats_dev || !edge_index_dev || !w_dev) return fail(c, MGU_ERR_INVALID, "mgu_ncut_edge_weights: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(ncut_edge_weight_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, s, feats_dev, D, edge_index_dev,
                     edge_index_dev + E, E, w_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_ncut_forward(mgu_ctx* c, const float* feats_dev, int N, int D, const int32_t* rowptr_src_dev, const int32_t* col_tgt_dev,
                     int64_t E, const float* assign_dev, int K, int assign_is_logits, float* soft_dev, int32_t* hard_dev,
                     float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (N <= 0 || D <= 0 || E < 0 || K < 1 || K > NCUT_MAX_K)
    return fail(c, MGU_ERR_INVALID, "mgu_ncut_forward: unsupported sizes N=%d D=%d E=%lld K=%d (1 <= K <= %d)", N, D, (long long)E, K,
                NCUT_MAX_K);
  if (!feats_dev || !rowptr_src_dev || (E > 0 && !col_tgt_dev) || !assign_dev || !loss_dev)
    return fail(c, MGU_ERR_INVALID, "mgu_ncut_forward: NULL buffer");
  if (assign_is_logits && !soft_dev) return fail(c, MGU_ERR_INVALID, "mgu_ncut_forward: soft_dev is required with logits");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  int rc = ensure(c, &c->ncws, &c->ncws_bytes, (size_t)NCUT_SLOTS * 2 * NCUT_MAX_K * sizeof(double));
  if (rc) return rc;
  double* acc = (double*)c->ncws;
  HIPCHK(c, hipMemsetAsync(acc, 0, (size_t)NCUT_SLOTS * 2 * K * sizeof(double), s));
  const float* P = assign_dev;
  if (assign_is_logits) {
    hipLaunchKernelGGL(ncut_softmax_kernel, dim3((N + 255) / 256), dim3(256), 0, s, assign_dev, N, K, soft_dev, hard_dev);
    HIPCHK(c, hipGetLastError());
    P = soft_dev;
  }
  if ((D & 3) == 0 && ((uintptr_t)feats_dev & 15) == 0)
    hipLaunchKernelGGL(ncut_node16_kernel, dim3((N + 15) / 16), dim3(256), 0, s, feats_dev, N, D, P, K, rowptr_src_dev, col_tgt_dev, acc);
  else
    hipLaunchKernelGGL(ncut_node_kernel, dim3((N + 3) / 4), dim3(256), 0, s, feats_dev, N, D, P, K, rowptr_src_dev, col_tgt_dev, acc);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(ncut_finalize_kernel, dim3(1), dim3(64), 0, s, acc, K, loss_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_ncut_backward(mgu_ctx* c, const float* feats_dev, int N, int D, const int32_t* rowptr_src_dev, const int32_t* col_tgt_dev,
                      const int32_t* rowptr_tgt_dev, const int32_t* col_src_dev, int64_t E, const float* soft_dev, int K,
                      int assign_is_logits, const float* gloss_dev, const float* gsoft_dev, float* dassign_dev, float* dfeats_dev,
                      void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (N <= 0 || D <= 0 || D > 64 * NCUT_BWD_DREGS || E < 0 || K < 1 || K > NCUT_MAX_K)
    return fail(c, MGU_ERR_INVALID, "mgu_ncut_backward: unsupported sizes N=%d D=%d E=%lld K=%d (D <= %d, 1 <= K <= %d)", N, D,
                (long long)E, K, 64 * NCUT_BWD_DREGS, NCUT_MAX_K);
  if (!feats_dev || !rowptr_src_dev || !rowptr_tgt_dev || (E > 0 && (!col_tgt_dev || !col_src_dev)) || !soft_dev || !dassign_dev)
    return fail(c, MGU_ERR_INVALID, "mgu_ncut_backward: NULL buffer");
  if (gsoft_dev && !assign_is_logits)
    return fail(c, MGU_ERR_INVALID, "mgu_ncut_backward: gsoft_dev only exists when the assignments were logits");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const size_t acc_bytes = (size_t)NCUT_SLOTS * 2 * NCUT_MAX_K * sizeof(double);
  int rc = ensure(c, &c->ncws, &c->ncws_bytes, acc_bytes + 2 * NCUT_MAX_K * sizeof(float));
  if (rc) return rc;
  double* acc = (double*)c->ncws;
  float* coef = (float*)((char*)c->ncws + acc_bytes);
  // cut_k / assoc_k again (the forward keeps no state between calls: its scratch is shared by every MinCut module of the ctx)
  HIPCHK(c, hipMemsetAsync(acc, 0, (size_t)NCUT_SLOTS * 2 * K * sizeof(double), s));
  if ((D & 3) == 0 && ((uintptr_t)feats_dev & 15) == 0)
    hipLaunchKernelGGL(ncut_node16_kernel, dim3((N + 15) / 16), dim3(256), 0, s, feats_dev, N, D, soft_dev, K, rowptr_src_dev, col_tgt_dev, acc);
  else
    hipLaunchKernelGGL(ncut_node_kernel, dim3((N + 3) / 4), dim3(256), 0, s, feats_dev, N, D, soft_dev, K, rowptr_src_dev, col_tgt_dev, acc);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(ncut_coef_kernel, dim3(1), dim3(64), 0, s, acc, K, gloss_dev, coef);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(ncut_bwd_node_kernel, dim3((N + 3) / 4), dim3(256), 0, s, feats_dev, N, D, soft_dev, K, rowptr_src_dev, col_tgt_dev,
                     rowptr_tgt_dev, col_src_dev, coef, gsoft_dev, assign_is_logits, dassign_dev, dfeats_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_relu_backward(mgu_ctx* c, const float* dy_dev, const float* y_dev, int64_t n, float* dz_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (n < 0) return fail(c, MGU_ERR_INVALID, "mgu_relu_backward: n=%lld", (long long)n);
  if (n == 0) return MGU_OK;
  if (!dy_dev || !y_dev || !dz_dev) return fail(c, MGU_ERR_INVALID, "mgu_relu_backward: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, dy_dev, y_dev, n, dz_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

}  // extern "C"
