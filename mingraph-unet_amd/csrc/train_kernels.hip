// HBM-bound kernels of the training step (scripts/train_segmentation.py:117-137): train-mode
// BatchNorm (batch statistics, running-stat update, backward), ReLU mask, MaxPool backward,
// softmax cross-entropy forward+backward, weight-panel packing for dgrad, gradient unpacking, Adam.
// All tensors NHWC fp32 with an explicit pixel pitch (ld) so channel slices of the concat buffers work.
#include "common.h"
#include "pack_small.h"

namespace mgu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int nblk(int64_t work, int threads, int cap = 256 * 8) {
  int64_t b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------
// Per-channel reductions over the M pixels of an NHWC tensor.  Thread (q = t % (C/4), pl = t / (C/4))
// accumulates float4 channel quads over rows pl, pl+npl, ... of the block's slab in fp32, the block
// combines through LDS and adds ONE double per channel into slot (blockIdx % RED_SLOTS) of a slotted
// accumulator [RED_SLOTS][2*C] (same-line double atomics serialise at ~12 ns each -- MI355X_MICROARCH.md
// "fanin" -- so the adds are spread over 64 x 2C addresses); slot_reduce_kernel then folds the slots.
//   mode 0: acc0 = sum z,               acc1 = sum z*z                    (BatchNorm batch statistics)
//   mode 1: g = dy * (y > 0); xh = (z - mean) * invstd;  acc0 = sum g, acc1 = sum g*xh   (BN backward)
//           y = relu(scale*z + shift) is NOT read back: its sign is recomputed from z with the forward's scale/shift
//           (one tensor read less in each of the two backward passes: they run at HBM speed, so bytes are time)
//   mode 2: acc0 = sum z (column sums: bias gradients)
//   mode 3: BN backward apply: dz = gamma*invstd*(g - sum_g/M - xh*sum_gx/M) written to `dz`, acc0 = sum dz
// ------------------------------------------------------------------------------------------------

template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ dy,
                                                          int lddy, const float* __restrict__ fsc, const float* __restrict__ fsh,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const double* __restrict__ sums,
                                                          float* __restrict__ dz, int64_t M, int C, int rows_per_block,
                                                          double* __restrict__ slots) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][npl][C]
  constexpr bool TWO = (MODE == 0 || MODE == 1);
  const int q = C >> 2, npl = 256 / q, t = threadIdx.x;
  const int cq = t % q, pl = t / q;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = min(M, r_begin + rows_per_block);
  if (pl < npl) {
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f}, k0 = mu, k1 = mu, k2 = mu, sc = is, sh = mu;
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mu[j] = mean[cq * 4 + j];
        is[j] = invstd[cq * 4 + j];
        sc[j] = fsc[cq * 4 + j];
        sh[j] = fsh[cq * 4 + j];
      }
    }
    if (MODE == 3) {
      const float invM = (float)(1.0 / (double)M);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        k0[j] = gamma[cq * 4 + j] * is[j];
        k1[j] = (float)sums[cq * 4 + j] * invM;        // sum_g / M
        k2[j] = (float)sums[C + cq * 4 + j] * invM;    // sum_gx / M
      }
    }
#pragma unroll 4   // few, long workgroups (one table row each): several loads of a thread in flight keep the stream at HBM speed
    for (int64_t r = r_begin + pl; r < r_end; r += npl) {
      const f32x4 zv = *reinterpret_cast<const f32x4*>(z + r * ldz + cq * 4);
      if (MODE == 0) {
        a0 += zv;
        a1 += zv * zv;
      } else if (MODE == 1 || MODE == 3) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + r * lddy + cq * 4);
        const f32x4 yv = zv * sc + sh;   // pre-ReLU output of the forward (bn_apply_relu_kernel's expression)
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = yv[j] > 0.f ? dv[j] : 0.f;
          const float xh = (zv[j] - mu[j]) * is[j];
          if (MODE == 1) {
            a0[j] += g;
            a1[j] += g * xh;
          } else {
            o[j] = k0[j] * (g - k1[j] - xh * k2[j]);
            a0[j] += o[j];
          }
        }
        if (MODE == 3) *reinterpret_cast<f32x4*>(dz + r * C + cq * 4) = o;
      } else {
        a0 += zv;
      }
    }
    *reinterpret_cast<f32x4*>(red + pl * C + cq * 4) = a0;
    if (TWO) *reinterpret_cast<f32x4*>(red + (npl + pl) * C + cq * 4) = a1;
  }
  __syncthreads();
  double* slot = slots + (size_t)blockIdx.x * 2 * C;   // a row of this workgroup's own (the launcher keeps the grid <= CHAN_REDUCE_ROWS)
  for (int c = t; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = 0; i < npl; ++i) {
      s0 += (double)red[i * C + c];
      if (TWO) s1 += (double)red[(npl + i) * C + c];
    }
    atomicAdd(slot + c, s0);
    if (TWO) atomicAdd(slot + C + c, s1);
  }
}

// outd[j] = sum over slots (j < n, slot pitch = pitch); optional fp32 copies: outf0[j] for j < n0, outf1[j-n0] beyond
// The slots it read are cleared again, so the NEXT reduction needs no memset (the workspace is zeroed once, at allocation).
// block = 32 columns x 32 row lanes: lane r adds rows r, r + 32, ... (coalesced 256-byte reads; the loads of a lane are all issued
// before the rows are cleared, so they overlap), the 32 partial sums meet in LDS in a fixed order
template <int NIT>
__device__ __forceinline__ double fold_rows(double* __restrict__ col, int nrows, int rl, size_t pitch) {
  double v[NIT];
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int k = rl + 32 * u;
    v[u] = k < nrows ? col[(size_t)k * pitch] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < NIT; ++u) s += v[u];
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int k = rl + 32 * u;
    if (k < nrows) col[(size_t)k * pitch] = 0.0;
  }
  return s;
}
constexpr int FOLD_NIT = (STAT_ROWS > CHAN_REDUCE_ROWS ? STAT_ROWS : CHAN_REDUCE_ROWS) / 32;   // rows <= 32 * FOLD_NIT

__global__ __launch_bounds__(1024) void slot_reduce_kernel(double* __restrict__ slots, int nrows, int n, int pitch, double* __restrict__ outd,
                                                           float* __restrict__ outf0, int n0, float* __restrict__ outf1) {
  __shared__ double part[32][33];
  const int e = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + e;
  double s = j < n ? fold_rows<FOLD_NIT>(slots + j, nrows, rl, (size_t)pitch) : 0.0;
  part[rl][e] = s;
  __syncthreads();
  if (rl != 0 || j >= n) return;
#pragma unroll
  for (int k = 1; k < 32; ++k) s += part[k][e];
  if (outd) outd[j] = s;
  if (j < n0) {
    if (outf0) outf0[j] = (float)s;
  } else if (outf1) {
    outf1[j - n0] = (float)s;
  }
}

size_t chan_reduce_work_bytes(int Cmax) {
  return (size_t)std::max(STAT_ROWS, CHAN_REDUCE_ROWS) * 2 * Cmax * sizeof(double);
}

// work: chan_reduce_work_bytes(C) of scratch.  Results: outd[0..n) doubles (n = 2C for modes 0/1, C for 2/3).
static hipError_t launch_chan_reduce(int mode, const float* z, int ldz, const float* dy, int lddy, const float* fsc, const float* fsh,
                                     const float* mean, const float* invstd, const float* gamma, const double* sums,
                                     float* dz, int64_t M, int C, double* work, double* outd, float* outf0, int n0,
                                     float* outf1, hipStream_t s, int* defer_rows = nullptr) {
  if ((C & 3) || C < 4 || C > 1024 || (ldz & 3) || M < 1) return hipErrorInvalidValue;
  int64_t blocks = (M + 63) / 64;   // `work` is zero on entry and on exit (slot_reduce_kernel cleans up)
  if (blocks > CHAN_REDUCE_ROWS) blocks = CHAN_REDUCE_ROWS;   // one row of the table per workgroup
  const int rows = (int)((M + blocks - 1) / blocks);
  blocks = (M + rows - 1) / rows;
  const int npl = 256 / (C >> 2);
  const size_t lds = (size_t)2 * npl * C * sizeof(float);
  dim3 g((unsigned)blocks), b(256);
#define MGU_CR(MODE) hipLaunchKernelGGL(chan_reduce_kernel<MODE>, g, b, lds, s, z, ldz, dy, lddy, fsc, fsh, mean, invstd, gamma, sums, dz, M, C, rows, work)
  if (mode == 0) MGU_CR(0);
  else if (mode == 1) MGU_CR(1);
  else if (mode == 2) MGU_CR(2);
  else MGU_CR(3);
#undef MGU_CR
  const int n = (mode == 0 || mode == 1) ? 2 * C : C;
  if (defer_rows) {   // the caller folds (and clears) the rows later, in a launch it makes anyway, before `work` is used again
    *defer_rows = (int)blocks;
    return hipGetLastError();
  }
  hipLaunchKernelGGL(slot_reduce_kernel, dim3((n + 31) / 32), dim3(1024), 0, s, work, (int)blocks, n, 2 * C, outd, outf0, n0, outf1);
  return hipGetLastError();
}

hipError_t launch_bn_stats(const float* z, int ldz, int64_t M, int C, double* work, double* sums, hipStream_t s) {
  return launch_chan_reduce(0, z, ldz, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, C, work, sums,
                            nullptr, 0, nullptr, s);
}
// sums[0..C) = sum g (= dbeta), sums[C..2C) = sum g*xhat (= dgamma); fp32 copies go straight to the flat gradient
hipError_t launch_bn_bwd_reduce(const float* dy, int lddy, const float* fsc, const float* fsh, const float* z, int ldz,
                                const float* mean, const float* invstd, int64_t M, int C, double* work, double* sums,
                                float* dbeta, float* dgamma, hipStream_t s) {
  return launch_chan_reduce(1, z, ldz, dy, lddy, fsc, fsh, mean, invstd, nullptr, nullptr, nullptr, M, C, work, sums, dbeta, C,
                            dgamma, s);
}
// dz (dense, pitch C) and its column sum (the conv bias gradient) in one pass
hipError_t launch_bn_bwd_apply(const float* dy, int lddy, const float* fsc, const float* fsh, const float* z, const float* mean,
                               const float* invstd, const float* gamma, const double* sums, int64_t M, int C, float* dz,
                               double* work, float* dbias, hipStream_t s) {
  return launch_chan_reduce(3, z, C, dy, lddy, fsc, fsh, mean, invstd, gamma, sums, dz, M, C, work, nullptr, dbias, C, nullptr, s);
}
// the same without the fold of the column sums: rows [0, *rows) of `work` (pitch 2C) hold them until launch_unpack_conv_grad(..., fold)
// folds them into the bias gradient -- one launch less per layer of the backward pass
hipError_t launch_bn_bwd_apply_deferred(const float* dy, int lddy, const float* fsc, const float* fsh, const float* z, const float* mean,
                                        const float* invstd, const float* gamma, const double* sums, int64_t M, int C, float* dz,
                                        double* work, int* rows, hipStream_t s) {
  return launch_chan_reduce(3, z, C, dy, lddy, fsc, fsh, mean, invstd, gamma, sums, dz, M, C, work, nullptr, nullptr, C, nullptr, s, rows);
}
hipError_t launch_colsum(const float* z, int ldz, int64_t M, int C, double* work, float* out, hipStream_t s) {
  return launch_chan_reduce(2, z, ldz, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, C, work, nullptr,
                            out, C, nullptr, s);
}

// ---- BatchNorm2d training forward, finalize (unet_encoder.py:12-13: eps 1e-5, momentum 0.1) -------------
// mean, biased var from the double sums; running stats updated IN PLACE with the unbiased variance.
__global__ void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq, double M, float eps,
                                   float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ run_mean, float* __restrict__ run_var, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const double mu = sum[i] / M;
  double var = sumsq[i] / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[i] = (float)mu;
  invstd[i] = is;
  const float sc = gamma[i] * is;
  scale[i] = sc;
  shift[i] = beta[i] - (float)mu * sc;
  if (run_mean) {
    const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
    run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * (float)mu;
    run_var[i] = (1.f - momentum) * run_var[i] + momentum * (float)unb;
  }
}

hipError_t launch_bn_finalize(const double* sum, const double* sumsq, int64_t M, float eps, float momentum,
                              const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                              float* shift, float* run_mean, float* run_var, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sum, sumsq, (double)M, eps, momentum,
                     gamma, beta, mean, invstd, scale, shift, run_mean, run_var, C);
  return hipGetLastError();
}

// ---- finalize straight from the slotted accumulator (statistics accumulated by the conv epilogue) -----------------
// Folds the nrows (= workgroups of the producing launch) rows of channel c (sum at [k][c], sum of squares at [k][C + c]) in a
// fixed order, clears them, and finishes BatchNorm2d exactly as bn_finalize_kernel does: one launch instead of reduction + row
// fold + finalize.  Block = 32 channels x 32 row lanes.
__global__ __launch_bounds__(1024) void bn_finalize_slots_kernel(double* __restrict__ slots, int nrows, double* __restrict__ sums, double M,
                                         float eps, float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                         float* __restrict__ shift, float* __restrict__ run_mean, float* __restrict__ run_var, int C) {
  __shared__ double part[2][32][33];
  const int e = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  double s0 = 0.0, s1 = 0.0;
  if (i < C) {
    s0 = fold_rows<FOLD_NIT>(slots + i, nrows, rl, (size_t)2 * C);
    s1 = fold_rows<FOLD_NIT>(slots + C + i, nrows, rl, (size_t)2 * C);
  }
  part[0][rl][e] = s0, part[1][rl][e] = s1;
  __syncthreads();
  if (rl != 0 || i >= C) return;
#pragma unroll
  for (int k = 1; k < 32; ++k) s0 += part[0][k][e], s1 += part[1][k][e];
  if (sums) sums[i] = s0, sums[C + i] = s1;
  const double mu = s0 / M;
  double var = s1 / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[i] = (float)mu;
  invstd[i] = is;
  const float sc = gamma[i] * is;
  scale[i] = sc;
  shift[i] = beta[i] - (float)mu * sc;
  if (run_mean) {
    const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
    run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * (float)mu;
    run_var[i] = (1.f - momentum) * run_var[i] + momentum * (float)unb;
  }
}

hipError_t launch_bn_finalize_slots(double* slots, int nrows, double* sums, int64_t M, float eps, float momentum, const float* gamma,
                                    const float* beta, float* mean, float* invstd, float* scale, float* shift, float* run_mean,
                                    float* run_var, int C, hipStream_t s) {
  if (nrows < 1 || nrows > STAT_ROWS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_finalize_slots_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, slots, nrows, sums, (double)M, eps, momentum, gamma, beta,
                     mean, invstd, scale, shift, run_mean, run_var, C);
  return hipGetLastError();
}

// ---- y = relu(scale*z + shift), output with pitch/offset ---------------------------------------------
__global__ void bn_apply_relu_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                     const float* __restrict__ shift, float* __restrict__ y, int ldy, int64_t M, int C) {
  const int q = C >> 2;
  const int64_t total = M * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c0 = (int)(i - r * q) * 4;
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + r * C + c0);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c0);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf(zv[j] * sc[j] + sh[j], 0.f);
    *reinterpret_cast<f32x4*>(y + r * ldy + c0) = o;
  }
}

// the same with MaxPool2d(2) (unet_encoder.py:48) of y in the same pass: a thread owns a 2 x 2 window of one channel quad
// (even H and W: every pixel belongs to exactly one window); saves the pool kernel's re-read of y
__global__ void bn_apply_relu_pool_kernel(const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                                          float* __restrict__ y, int ldy, float* __restrict__ pooled, int B, int H, int W, int C) {
  const int q = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % q) * 4;
    int64_t r = i / q;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), n = (int)(r / Ho);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c0);
    f32x4 m = {0.f, 0.f, 0.f, 0.f};   // ReLU outputs are >= 0
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int64_t pix = ((int64_t)n * H + 2 * yo + dy) * W + 2 * xo + dx;
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + pix * C + c0);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(zv[j] * sc[j] + sh[j], 0.f), m[j] = fmaxf(m[j], o[j]);
        *reinterpret_cast<f32x4*>(y + pix * ldy + c0) = o;
      }
    *reinterpret_cast<f32x4*>(pooled + (((int64_t)n * Ho + yo) * Wo + xo) * C + c0) = m;
  }
}

hipError_t launch_bn_apply_relu_pool(const float* z, const float* scale, const float* shift, float* y, int ldy, float* pooled, int B, int H,
                                     int W, int C, hipStream_t s) {
  if ((H & 1) || (W & 1) || (C & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_apply_relu_pool_kernel, dim3(nblk((int64_t)B * (H >> 1) * (W >> 1) * (C >> 2), 256)), dim3(256), 0, s, z, scale,
                     shift, y, ldy, pooled, B, H, W, C);
  return hipGetLastError();
}

hipError_t launch_bn_apply_relu(const float* z, const float* scale, const float* shift, float* y, int ldy, int64_t M, int C,
                                hipStream_t s) {
  hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(nblk(M * (C >> 2), 256)), dim3(256), 0, s, z, scale, shift, y, ldy, M, C);
  return hipGetLastError();
}

// ---- MaxPool2d(2,2) backward, accumulated into dskip (first maximum wins, like aten) ------------------
__global__ void maxpool2_bwd_add_kernel(const float* __restrict__ ypre, int ldy, const float* __restrict__ dpool,
                                        float* __restrict__ dskip, int ldd, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, q = C >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % q) * 4;
    int64_t pix = i / q;
    const int xo = (int)(pix % Wo);
    pix /= Wo;
    const int yo = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int64_t p00 = ((int64_t)n * H + 2 * yo) * W + 2 * xo;
    const int64_t offs[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(ypre + offs[k] * ldy + c0);
    const f32x4 dp = *reinterpret_cast<const f32x4*>(dpool + (((int64_t)n * Ho + yo) * Wo + xo) * C + c0);
    f32x4 add[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int best = 0;
      float bv = v[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][j] > bv) { bv = v[k][j]; best = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) add[k][j] = (k == best) ? dp[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4* p = reinterpret_cast<f32x4*>(dskip + offs[k] * ldd + c0);
      *p = *p + add[k];
    }
  }
}

hipError_t launch_maxpool2_bwd_add(const float* y, int ldy, const float* dpool, float* dskip, int ldd, int B, int H, int W,
                                   int C, hipStream_t s) {
  const int64_t total = (int64_t)B * (H >> 1) * (W >> 1) * (C >> 2);
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(maxpool2_bwd_add_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, y, ldy, dpool, dskip, ldd, B, H, W, C);
  return hipGetLastError();
}

// ---- zero channels [coff, coff+C) of the pixels outside the h2 x w2 top-left region (F.pad backward) ----
__global__ void zero_pad_region_kernel(float* __restrict__ buf, int ld, int coff, int C, int B, int H, int W, int h2, int w2) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    if (y >= h2 || x >= w2)
      for (int c = 0; c < C; ++c) buf[i * ld + coff + c] = 0.f;
  }
}
hipError_t launch_zero_pad_region(float* buf, int ld, int coff, int C, int B, int H, int W, int h2, int w2, hipStream_t s) {
  hipLaunchKernelGGL(zero_pad_region_kernel, dim3(nblk((int64_t)B * H * W, 256)), dim3(256), 0, s, buf, ld, coff, C, B, H, W, h2, w2);
  return hipGetLastError();
}

// ---- nn.CrossEntropyLoss (mean, ignore_index = -100) forward + gradient w.r.t. logits (train_segmentation.py:91,127) ----
// logits NHWC (M, C); labels int64 (M); dlogits written with pitch ldd >= C (pad columns zeroed).
// torch semantics: a pixel whose label is ignore_index contributes nothing to the loss, gets a zero gradient and is left
// out of the mean's denominator; any other label outside [0, C) is an error (torch raises / device-asserts): here it is
// never used as an index, the pixel gets a zero gradient, the loss becomes NaN and *err_word (host-visible) is set.
//   acc[0] = sum_i -log p_i[y_i] (double), acc[1] = number of counted pixels (double)
__global__ __launch_bounds__(256) void ce_count_kernel(const int64_t* __restrict__ labels, int64_t M, int C, long long ignore_index,
                                                       double* __restrict__ acc, int* __restrict__ err_word) {
  __shared__ int red[256];
  int local = 0, bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const long long y = labels[i];
    if (y == ignore_index) continue;
    if (y < 0 || y >= C) bad = 1;
    else ++local;
  }
  if (bad && err_word) atomicOr(err_word, 1);
  red[threadIdx.x] = local;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0]) atomicAdd(acc + 1, (double)red[0]);
}

__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         int64_t M, int C, long long ignore_index, float grad_scale,
                                                         float* __restrict__ dlogits, int ldd, double* __restrict__ acc) {
  __shared__ double red[256];
  double local = 0.0;
  // grad_scale = 1/M reproduces loss.backward() of the mean loss: with ignored pixels the mean runs over `count`
  const double count = acc[1];
  const float gs = count > 0.0 ? (float)((double)grad_scale * (double)M / count) : 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = logits + i * C;
    const long long yl = labels[i];
    const bool counted = yl >= 0 && yl < C;        // ignore_index and invalid labels: zero gradient
    const int y = counted ? (int)yl : 0;
    float mx = p[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
    const float lse = mx + logf(se);
    if (counted) local += (double)(lse - p[y]);
    else if (yl != ignore_index) local += (double)__builtin_nanf("");   // out-of-range label: the loss says so
    const float inv = 1.f / se;
    for (int c = 0; c < ldd; ++c) {
      float g = 0.f;
      if (c < C && counted) g = (expf(p[c] - mx) * inv - (c == y ? 1.f : 0.f)) * gs;
      dlogits[i * ldd + c] = g;
    }
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}

// mean over the counted pixels; 0 counted pixels -> NaN, as torch's 0/0
__global__ void ce_finalize_kernel(const double* acc, float* loss_out) { *loss_out = (float)(acc[0] / acc[1]); }

hipError_t launch_ce(const float* logits, const int64_t* labels, int64_t M, int C, long long ignore_index, float grad_scale,
                     float* dlogits, int ldd, double* acc, int* err_word, float* loss_out, hipStream_t s) {
  hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(double), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ce_count_kernel, dim3(nblk(M, 256, 1024)), dim3(256), 0, s, labels, M, C, ignore_index, acc, err_word);
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(nblk(M, 256, 2048)), dim3(256), 0, s, logits, labels, M, C, ignore_index, grad_scale,
                     dlogits, ldd, acc);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1), 0, s, acc, loss_out);
  return hipGetLastError();
}

// ---- dgrad weight panels -----------------------------------------------------------------------------
// conv3x3 / 1x1: din = conv(dz, W') with W'[ci][(2-r,2-s), co] = W[co][ci][r][s]:
// panel [Cin][Kp], k = tap'*Cop + co  (Cop = Cout rounded up to 4, zero padded)
__global__ void pack_dgrad_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int Cop, int KS,
                                    int Kp) {
  pack_dgrad_w_body(w, wp, Cout, Cin, Cop, KS, Kp, blockIdx.x, gridDim.x);
}
hipError_t launch_pack_dgrad_w(const float* w, float* wp, int Cout, int Cin, int Cop, int KS, int Kp, hipStream_t s) {
  hipLaunchKernelGGL(pack_dgrad_w_kernel, dim3(nblk((int64_t)Cin * Kp, 256)), dim3(256), 0, s, w, wp, Cout, Cin, Cop, KS, Kp);
  return hipGetLastError();
}
// ConvTranspose2d dgrad: dprev[m][ci] = sum_{q,co} dup[pix(m,q)][co] * W[ci][co][q]: panel [Cin][Kp], k = q*Cout + co
__global__ void pack_convt_dgrad_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int Kp) {
  const int64_t total = (int64_t)Cin * Kp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i / Kp), k = (int)(i - (int64_t)ci * Kp);
    const int q = k / Cout, co = k - q * Cout;
    wp[i] = (q < 4) ? w[((int64_t)ci * Cout + co) * 4 + q] : 0.f;
  }
}
hipError_t launch_pack_convt_dgrad_w(const float* w, float* wp, int Cin, int Cout, int Kp, hipStream_t s) {
  hipLaunchKernelGGL(pack_convt_dgrad_w_kernel, dim3(nblk((int64_t)Cin * Kp, 256)), dim3(256), 0, s, w, wp, Cin, Cout, Kp);
  return hipGetLastError();
}

// ---- gradient panels -> the reference's parameter layouts ---------------------------------------------
// conv: panel [Cout][Kp], k = tap*Cp + c  ->  OIHW (Cout, Cin, KS, KS)
// Sums `groups` partial panels in a fixed order (the atomics-free wgrad path writes one panel per patch group) and writes OIHW.
// A workgroup owns one output channel and 32 consecutive input channels with all KS*KS taps.  Thread = (group slice sl of 32, channel
// quad cq of 8): per tap ONE 16-byte load of every partial panel of its slice (8 lanes = the 128-byte row piece; the scalar-load
// version moved 128 bytes per instruction and ran at 2.7 TB/s), the 32 slice sums meet in LDS, and the OIHW destination of those
// 32 x TAPS values is ONE contiguous run written in order (a thread per panel element wrote 4 bytes at a 36-byte stride: 56 us for
// the 512 x 512 layer, whose reads take 10).
// Workgroups past `ublocks` (fold.slots != nullptr) fold the deferred column sums of the layer's BatchNorm backward instead
// (launch_bn_bwd_apply_deferred): 32 columns per workgroup, row lane rl adds rows rl, rl + 8, ... and clears them, the eight partial
// sums meet in LDS in a fixed order.
struct SlotFold {
  double* slots;   // nullptr: nothing to fold
  int nrows, n, pitch;
  float* out;
};
template <int TAPS>
__global__ __launch_bounds__(256) void unpack_conv_grad_kernel(const float* __restrict__ dwp, int groups, size_t panel_stride,
                                                               float* __restrict__ g, int Cout, int Cin, int Cp, int Kp, int ublocks,
                                                               SlotFold fold) {
  __shared__ __attribute__((aligned(16))) float part[32][TAPS][33];
  if ((int)blockIdx.x >= ublocks) {   // block-uniform
    double* fpart = reinterpret_cast<double*>(&part[0][0][0]);   // [8][33] doubles = 2112 bytes (part has >= 4224)
    const int e = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int j = ((int)blockIdx.x - ublocks) * 32 + e;
    double sacc = 0.0;
    if (j < fold.n) {
      double* col = fold.slots + j;
      for (int k0 = rl; k0 < fold.nrows; k0 += 64) {   // eight independent loads in flight
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k0 + 8 * u < fold.nrows ? col[(size_t)(k0 + 8 * u) * fold.pitch] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += v[u];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 8 * u < fold.nrows) col[(size_t)(k0 + 8 * u) * fold.pitch] = 0.0;
      }
    }
    fpart[rl * 33 + e] = sacc;
    __syncthreads();
    if (rl == 0 && j < fold.n) {
#pragma unroll
      for (int k = 1; k < 8; ++k) sacc += fpart[k * 33 + e];
      fold.out[j] = (float)sacc;
    }
    return;
  }
  const int cq = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int nchunk = (Cp + 31) / 32;
  for (int item = blockIdx.x; item < Cout * nchunk; item += ublocks) {
    const int co = item / nchunk, ci0 = (item - co * nchunk) * 32;
    const bool ok = ci0 + cq * 4 < Cp;   // Cp is a multiple of 4: a quad is whole or absent
    const float* p = dwp + (int64_t)co * Kp + (ok ? ci0 + cq * 4 : 0);
    f32x4 s[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = sl; q < groups; q += 32) {   // TAPS independent 16-byte loads in flight
      const float* pq = p + (size_t)q * panel_stride;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) s[t] += *reinterpret_cast<const f32x4*>(pq + t * Cp);
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[sl][t][cq * 4 + e] = ok ? s[t][e] : 0.f;
    __syncthreads();
    const int nci = min(32, Cin - ci0);   // channels of this chunk that exist in the OIHW tensor (Cp pads to a multiple of 4)
    for (int o = threadIdx.x; o < nci * TAPS; o += 256) {
      const int cl = o / TAPS, t = o - cl * TAPS;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) sum += part[j][t][cl];
      g[((int64_t)co * Cin + ci0) * TAPS + o] = sum;
    }
    __syncthreads();
  }
}
// fold_rows > 0: also folds rows [0, fold_rows) of the reduction slots (pitch 2 fold_n) into fold_out[0 .. fold_n) and clears them
hipError_t launch_unpack_conv_grad(const float* dwp, int groups, size_t panel_stride, float* g, int Cout, int Cin, int Cp, int KS,
                                   int Kp, hipStream_t s, double* fold_slots, int fold_rows, int fold_n, float* fold_out) {
  int64_t blocks = (int64_t)Cout * ((Cp + 31) / 32);
  if (blocks > 65535) blocks = 65535;
  SlotFold f{nullptr, 0, 0, 0, nullptr};
  int fblocks = 0;
  if (fold_slots && fold_rows > 0 && fold_n > 0) f = SlotFold{fold_slots, fold_rows, fold_n, 2 * fold_n, fold_out}, fblocks = (fold_n + 31) / 32;
  if (KS == 3)
    hipLaunchKernelGGL(unpack_conv_grad_kernel<9>, dim3((unsigned)blocks + fblocks), dim3(256), 0, s, dwp, groups, panel_stride, g, Cout, Cin, Cp, Kp,
                       (int)blocks, f);
  else
    hipLaunchKernelGGL(unpack_conv_grad_kernel<1>, dim3((unsigned)blocks + fblocks), dim3(256), 0, s, dwp, groups, panel_stride, g, Cout, Cin, Cp, Kp,
                       (int)blocks, f);
  return hipGetLastError();
}
// convT: partial panels [groups][Cin][Kp], k = q*Cout + co  ->  (Cin, Cout, 2, 2), the panels added in a fixed order.
// Same shape as unpack_conv_grad_kernel: a workgroup owns 32 consecutive panel elements (128-byte coalesced reads of every
// partial panel) and splits the group loop over 8 slices with four loads in flight each -- the thin ConvTranspose layers write
// ~1000 partial panels (pixel splits are their only parallelism), and a thread per element summing them one after the other
// took 123 us on the shallowest layer.
__global__ __launch_bounds__(256) void unpack_convt_grad_kernel(const float* __restrict__ dwp, int groups, size_t panel_stride,
                                                                float* __restrict__ g, int Cin, int Cout, int Kp) {
  __shared__ float part[8][32];
  const int K = 4 * Cout;
  const int64_t total = (int64_t)Cin * K;
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (int64_t i0 = (int64_t)blockIdx.x * 32; i0 < total; i0 += (int64_t)gridDim.x * 32) {
    const int64_t i = i0 + e;
    const bool ok = i < total;
    const int k = ok ? (int)(i % K) : 0, ci = ok ? (int)(i / K) : 0;
    const float* p = dwp + (int64_t)ci * Kp + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = sl;
    for (; q + 24 < groups; q += 32) {   // 4 independent loads in flight
      s0 += p[(size_t)q * panel_stride];
      s1 += p[(size_t)(q + 8) * panel_stride];
      s2 += p[(size_t)(q + 16) * panel_stride];
      s3 += p[(size_t)(q + 24) * panel_stride];
    }
    for (; q < groups; q += 8) s0 += p[(size_t)q * panel_stride];
    part[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && ok) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += part[j][e];
      const int tap = k / Cout, co = k - tap * Cout;
      g[((int64_t)ci * Cout + co) * 4 + tap] = sum;
    }
    __syncthreads();
  }
}
hipError_t launch_unpack_convt_grad(const float* dwp, int groups, size_t panel_stride, float* g, int Cin, int Cout, int Kp, hipStream_t s) {
  int64_t blocks = ((int64_t)Cin * Cout * 4 + 31) / 32;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(unpack_convt_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dwp, groups, panel_stride, g, Cin, Cout, Kp);
  return hipGetLastError();
}

// ---- torch.optim.Adam (L2 weight decay folded into the gradient), train_segmentation.py:96 --------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                            float grad_scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] * grad_scale + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                       float wd, int step, float grad_scale, hipStream_t s) {
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2),
                     grad_scale);
  return hipGetLastError();
}

// torch.optim.SGD(lr, momentum, weight_decay) on the flat buffer (scripts/train_segmentation.py:97-98; dampening 0, no Nesterov):
//   g = grad_scale * grad + wd * p;  buf = step == 1 ? g : momentum * buf + g  (momentum != 0);  p -= lr * (momentum ? buf : g)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n, float lr,
                           float momentum, float wd, int first, float grad_scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i];
    float d = g[i] * grad_scale + wd * pi;
    if (momentum != 0.f) {
      d = first ? d : momentum * buf[i] + d;
      buf[i] = d;
    }
    p[i] = pi - lr * d;
  }
}
hipError_t launch_sgd(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd, int step, float grad_scale,
                      hipStream_t s) {
  hipLaunchKernelGGL(sgd_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, p, g, buf, n, lr, momentum, wd, step == 1 ? 1 : 0, grad_scale);
  return hipGetLastError();
}

}  // namespace mgu
