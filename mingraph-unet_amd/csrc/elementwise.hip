// HBM-bound helper kernels around the implicit-GEMM: layout packing, 2x2 max-pool, patch mean,
// weight repacking, BatchNorm folding, class argmax.  All NHWC, 16-byte lanes where channels allow.
#include "common.h"
#include "pack_small.h"

namespace mgu {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 16-byte chunk <-> fp32 lanes for the two storage types (fp32: 4 elements, bf16: 8 elements)
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void load(const float* p, float* f) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0], f[1] = v[1], f[2] = v[2], f[3] = v[3];
  }
  static __device__ __forceinline__ void store(float* p, const float* f) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  }
};
template <> struct Chunk<__bf16> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void load(const __bf16* p, float* f) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
  }
  static __device__ __forceinline__ void store(__bf16* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
};

static inline int nblocks(int64_t work, int threads, int cap = 256 * 16) {
  int64_t b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ---- input: arbitrary-stride (n,c,y,x) fp32 -> NHWC with channels zero padded to Cp (Cp % 4 == 0) ----
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, T* __restrict__ out, int64_t npix_total, int C, int Cp, int H,
                                  int W, int64_t sn, int64_t sc, int64_t sh, int64_t sw) {
  constexpr int VEC = Chunk<T>::VEC;
  const int q = Cp / VEC;
  const int64_t total = npix_total * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / q;
    const int c0 = (int)(i - pix * q) * VEC;
    const int64_t n = pix / ((int64_t)H * W);
    const int64_t rem = pix - n * (int64_t)H * W;
    const int y = (int)(rem / W);
    const int xx = (int)(rem - (int64_t)y * W);
    const float* p = x + n * sn + y * sh + xx * sw;
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = (c0 + j < C) ? p[(c0 + j) * sc] : 0.f;
    Chunk<T>::store(out + pix * Cp + c0, v);
  }
}

hipError_t launch_pack_input(const float* x, void* out, int dtype, int B, int C, int Cp, int H, int W, int64_t sn, int64_t sc,
                             int64_t sh, int64_t sw, hipStream_t s) {
  const int64_t npix = (int64_t)B * H * W;
  if (dtype == 0)
    hipLaunchKernelGGL(pack_input_kernel<float>, dim3(nblocks(npix * (Cp >> 2), 256)), dim3(256), 0, s, x, (float*)out, npix, C,
                       Cp, H, W, sn, sc, sh, sw);
  else
    hipLaunchKernelGGL(pack_input_kernel<__bf16>, dim3(nblocks(npix * (Cp >> 3), 256)), dim3(256), 0, s, x, (__bf16*)out, npix,
                       C, Cp, H, W, sn, sc, sh, sw);
  return hipGetLastError();
}

// ---- MaxPool2d(2,2), floor mode (model/unet/unet_encoder.py:48,70); input may be a channel slice ----
template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ in, int ldin, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int VEC = Chunk<T>::VEC;
  const int Ho = H >> 1, Wo = W >> 1, q = C / VEC;
  const int64_t total = (int64_t)B * Ho * Wo * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % q) * VEC;
    int64_t pix = i / q;
    const int xo = (int)(pix % Wo);
    pix /= Wo;
    const int yo = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const T* p = in + (((int64_t)n * H + 2 * yo) * W + 2 * xo) * ldin + c0;
    float a[VEC], b[VEC], c[VEC], e[VEC], v[VEC];
    Chunk<T>::load(p, a);
    Chunk<T>::load(p + ldin, b);
    Chunk<T>::load(p + (int64_t)W * ldin, c);
    Chunk<T>::load(p + (int64_t)W * ldin + ldin, e);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], e[j]));
    Chunk<T>::store(out + (((int64_t)n * Ho + yo) * Wo + xo) * C + c0, v);
  }
}

hipError_t launch_maxpool2(const void* in, int ldin, void* out, int dtype, int B, int H, int W, int C, hipStream_t s) {
  const int vec = dtype == 0 ? 4 : 8;
  const int64_t total = (int64_t)B * (H >> 1) * (W >> 1) * (C / vec);
  if (total == 0) return hipSuccess;
  if (dtype == 0)
    hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(nblocks(total, 256)), dim3(256), 0, s, (const float*)in, ldin, (float*)out, B,
                       H, W, C);
  else
    hipLaunchKernelGGL(maxpool2_kernel<__bf16>, dim3(nblocks(total, 256)), dim3(256), 0, s, (const __bf16*)in, ldin, (__bf16*)out,
                       B, H, W, C);
  return hipGetLastError();
}

// ---- patch mean: one workgroup per patch; (Np, C) = mean over patch x patch window (zero padded) ----
// NC > 0: the same pass also applies the final 1x1 conv (unet_decoder.py:117,143) to every pixel it reads -- the
// decoder feature is 268 MB at 8 x 512^2 and both consumers are pure bandwidth, so reading it once instead of twice
// is the whole optimisation.  The C/VEC threads that hold a pixel's channels fold their partial dot products with a
// shuffle tree (C/VEC is a power of two <= 32 on this path).
template <typename T, int NC>
__global__ __launch_bounds__(256) void patch_mean_kernel(const T* __restrict__ feat, float* __restrict__ out, int H, int W,
                                                         int C, int patch, int nph, int npw, const float* __restrict__ hw,
                                                         const float* __restrict__ hb, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [npl][C]
  constexpr int VEC = Chunk<T>::VEC;
  const int q = C / VEC;
  const int npl = 256 / q;  // pixel lanes
  const int t = threadIdx.x;
  const int cq = t % q, pl = t / q;
  const int node = blockIdx.x;
  const int img = node / (nph * npw);
  const int pr = (node / npw) % nph, pc = node % npw;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  float wq[NC > 0 ? NC : 1][VEC];
  if (NC > 0) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
      for (int j = 0; j < VEC; ++j) wq[k][j] = hw[k * C + cq * VEC + j];
  }
  if (pl < npl) {
    // four pixels of a thread per trip, their loads issued together and UNCONDITIONALLY (a pixel outside the image or past the patch
    // reads the image's first pixel and is not used): with one dependent 16-byte load per trip the pass ran at 4.4 TB/s -- 32 waves
    // x 1 KB in flight per CU is what an HBM round trip of ~2 us sustains
    constexpr int UN = 4;
    const int64_t img_pix0 = (int64_t)img * H * W;
    for (int i0 = pl; i0 < patch * patch; i0 += UN * npl) {
      float v[UN][VEC];
      int64_t pix[UN];
      bool ok[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + u * npl;
        const int y = pr * patch + i / patch, x = pc * patch + i % patch;
        ok[u] = i < patch * patch && y < H && x < W;   // uniform over the q threads of a pixel
        pix[u] = ok[u] ? img_pix0 + (int64_t)y * W + x : img_pix0;
        Chunk<T>::load(feat + pix[u] * C + cq * VEC, v[u]);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += v[u][j];
        if (NC > 0) {
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            float dsum = 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) dsum += v[u][j] * wq[k][j];
            for (int o = 1; o < q; o <<= 1) dsum += __shfl_xor(dsum, o);
            if (cq == 0) logits[pix[u] * NC + k] = dsum + hb[k];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[pl * C + cq * VEC + j] = acc[j];
  }
  __syncthreads();
  if (t < C) {
    float sum = 0.f;
    for (int i = 0; i < npl; ++i) sum += red[i * C + t];
    out[(int64_t)node * C + t] = sum / (float)(patch * patch);
  }
}

// ncls > 0: also write logits[pixel][ncls] = head_w (ncls, C) . feat[pixel] + head_b (the fused 1x1 head); needs
// ncls <= 4 and C / vec a power of two <= 32 (head_fusable below)
bool patch_mean_head_fusable(int dtype, int C, int ncls) {
  const int vec = dtype == 0 ? 4 : 8;
  if (C % vec || ncls < 1 || ncls > 4) return false;
  const int q = C / vec;
  return q >= 1 && q <= 32 && (q & (q - 1)) == 0;
}

hipError_t launch_patch_mean(const void* feat, int dtype, float* out, int B, int H, int W, int C, int patch, hipStream_t s,
                             const float* head_w, const float* head_b, float* logits, int ncls) {
  const int vec = dtype == 0 ? 4 : 8;
  if ((C % vec) || C > 256 || C < vec) return hipErrorInvalidValue;
  if (ncls > 0 && (!patch_mean_head_fusable(dtype, C, ncls) || !head_w || !head_b || !logits)) return hipErrorInvalidValue;
  const int nph = (H + patch - 1) / patch, npw = (W + patch - 1) / patch;
  const int npl = 256 / (C / vec);
  const size_t lds = (size_t)npl * C * sizeof(float);
  const dim3 grid(B * nph * npw), block(256);
#define MGU_PM(T, NC) hipLaunchKernelGGL((patch_mean_kernel<T, NC>), grid, block, lds, s, (const T*)feat, out, H, W, C, patch, nph, npw, head_w, head_b, logits)
#define MGU_PM_T(T)             \
  do {                          \
    if (ncls <= 0) MGU_PM(T, 0); \
    else if (ncls == 1) MGU_PM(T, 1); \
    else if (ncls == 2) MGU_PM(T, 2); \
    else if (ncls == 3) MGU_PM(T, 3); \
    else MGU_PM(T, 4);          \
  } while (0)
  if (dtype == 0) MGU_PM_T(float);
  else MGU_PM_T(__bf16);
#undef MGU_PM_T
#undef MGU_PM
  return hipGetLastError();
}

// ---- weights: OIHW (Cout,Cin,KS,KS) -> panel [Cout][Kp], k = (r*KS+s)*Cp + c (zero padded) ------------
template <typename T>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cout, int Cin, int Cp, int KS, int Kp) {
  const int64_t total = (int64_t)Cout * Kp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / Kp), k = (int)(i - (int64_t)n * Kp);
    const int tap = k / Cp, c = k - tap * Cp;
    float v = 0.f;
    if (tap < KS * KS && c < Cin) v = w[((int64_t)n * Cin + c) * KS * KS + tap];
    wp[i] = (T)v;
  }
}

hipError_t launch_pack_conv_w(const float* w, void* wp, int dtype, int Cout, int Cin, int Cp, int KS, int Kp, hipStream_t s) {
  dim3 g(nblocks((int64_t)Cout * Kp, 256)), b(256);
  if (dtype == 0) hipLaunchKernelGGL(pack_conv_w_kernel<float>, g, b, 0, s, w, (float*)wp, Cout, Cin, Cp, KS, Kp);
  else hipLaunchKernelGGL(pack_conv_w_kernel<__bf16>, g, b, 0, s, w, (__bf16*)wp, Cout, Cin, Cp, KS, Kp);
  return hipGetLastError();
}

// ---- ConvTranspose2d weight (Cin,Cout,2,2) -> panel [(dy*2+dx)*Cout + co][Kp], k = ci ----------------
template <typename T>
__global__ void pack_convt_w_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int Kp) {
  const int64_t total = (int64_t)4 * Cout * Kp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / Kp), k = (int)(i - (int64_t)n * Kp);
    const int q = n / Cout, co = n - q * Cout;
    wp[i] = (T)((k < Cin) ? w[((int64_t)k * Cout + co) * 4 + q] : 0.f);
  }
}

hipError_t launch_pack_convt_w(const float* w, void* wp, int dtype, int Cin, int Cout, int Kp, hipStream_t s) {
  dim3 g(nblocks((int64_t)4 * Cout * Kp, 256)), b(256);
  if (dtype == 0) hipLaunchKernelGGL(pack_convt_w_kernel<float>, g, b, 0, s, w, (float*)wp, Cin, Cout, Kp);
  else hipLaunchKernelGGL(pack_convt_w_kernel<__bf16>, g, b, 0, s, w, (__bf16*)wp, Cin, Cout, Kp);
  return hipGetLastError();
}

// ---- eval BatchNorm + conv bias -> y = scale*acc + shift (unet_encoder.py:12-13,17-24) ----------------
__global__ void bn_fold_kernel(const float* bias, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, float* scale, float* shift, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = (bias[i] - mean[i]) * sc + beta[i];
  }
}

hipError_t launch_bn_fold(const float* bias, const float* gamma, const float* beta, const float* mean, const float* var,
                          float eps, float* scale, float* shift, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, s, bias, gamma, beta, mean, var, eps, scale,
                     shift, C);
  return hipGetLastError();
}

__global__ void bias_tile_kernel(const float* bias, float* shift, int C, int reps) { bias_tile_body(bias, shift, C, reps, blockIdx.x, gridDim.x); }

hipError_t launch_bias_tile(const float* bias, float* shift, int C, int reps, hipStream_t s) {
  hipLaunchKernelGGL(bias_tile_kernel, dim3((C * reps + 255) / 256), dim3(256), 0, s, bias, shift, C, reps);
  return hipGetLastError();
}

// ---- 1x1 convolution with a handful of output channels (the segmentation head, unet_decoder.py:117,143) ----
// HBM-bound (0.9 FLOP/B): 8 lanes read one pixel's channels as coalesced float4, each lane accumulates its partial
// dot products for up to 4 classes, an 8-lane shuffle tree folds them, lane 0 of the group stores.
template <typename T, int NC>
__global__ __launch_bounds__(256) void conv1x1_head_kernel(const T* __restrict__ in, int ldin, int C,
                                                           const float* __restrict__ w /*[NC][C]*/,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int ldout, int64_t npix) {
  const int lane8 = threadIdx.x & 7;
  const int64_t stride = (int64_t)gridDim.x * 32;
  for (int64_t p = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); p < npix; p += stride) {   // uniform trip count per 8-lane group
    float acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.f;
    constexpr int VEC = Chunk<T>::VEC;
    for (int c = lane8 * VEC; c < C; c += 8 * VEC) {
      float v[VEC];
      Chunk<T>::load(in + p * ldin + c, v);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[k] += v[j] * w[k * C + c + j];
      }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      acc[k] += __shfl_xor(acc[k], 1);
      acc[k] += __shfl_xor(acc[k], 2);
      acc[k] += __shfl_xor(acc[k], 4);
    }
    if (lane8 == 0) {
#pragma unroll
      for (int k = 0; k < NC; ++k) out[p * ldout + k] = acc[k] + (bias ? bias[k] : 0.f);
    }
  }
}

hipError_t launch_conv1x1_head(const void* in, int dtype, int ldin, int C, const float* w, const float* bias, float* out,
                               int ldout, int ncls, int64_t npix, hipStream_t s) {
  const int vec = dtype == 0 ? 4 : 8;
  if ((C % vec) || (ldin % vec) || ncls < 1 || ncls > 4) return hipErrorInvalidValue;
  const int blocks = nblocks(npix * 8, 256, 256 * 32);
#define MGU_HEAD(NC)                                                                                                         \
  do {                                                                                                                       \
    if (dtype == 0)                                                                                                          \
      hipLaunchKernelGGL((conv1x1_head_kernel<float, NC>), dim3(blocks), dim3(256), 0, s, (const float*)in, ldin, C, w, bias, out, \
                         ldout, npix);                                                                                       \
    else                                                                                                                     \
      hipLaunchKernelGGL((conv1x1_head_kernel<__bf16, NC>), dim3(blocks), dim3(256), 0, s, (const __bf16*)in, ldin, C, w, bias,   \
                         out, ldout, npix);                                                                                  \
  } while (0)
  if (ncls == 1) MGU_HEAD(1);
  else if (ncls == 2) MGU_HEAD(2);
  else if (ncls == 3) MGU_HEAD(3);
  else MGU_HEAD(4);
#undef MGU_HEAD
  return hipGetLastError();
}

// ---- first convolution: Conv2d(in_channels <= 4 -> NCO, k3, p1) on the packed NHWC4 input -----------------------
// K = 27..36 is far too short for an MFMA tile pipeline (the implicit-GEMM kernel spent its time in gather setup and
// epilogue: 146 us at 8 x 512^2, against ~65 us for the 302 MB it has to move).  One thread owns one pixel and all NCO
// output channels: the nine taps are nine coalesced 16-byte loads, the weights are wave-uniform, so they arrive through
// the scalar cache and every multiply-add is a VALU op with an SGPR operand -- no LDS, no barrier; the 128-byte channel
// vector of the pixel is stored as NCO/4 16-byte pieces.  HBM-bound.
// T = float: NHWC4 fp32 in, fp32 out.  T = __bf16 (bf16 storage mode): NHWC8 bf16 in (pack_input pads the 3 channels to one
// 16-byte pixel), bf16 out (8 channels per 16-byte store); the weights stay fp32 scalars either way.
template <typename T, int NCO, int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const T* __restrict__ in, const float* __restrict__ wf /*[9][4][NCO]*/,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            T* __restrict__ out, int64_t npix, int H, int W, int ldout, int coff,
                                                            int relu) {
  constexpr int LDI = sizeof(T) == 4 ? 4 : 8;   // elements per input pixel
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < npix;   // (lanes past the end still take part in the wave's LDS transpose)
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  float acc[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
  const T* base = in + p * LDI;
  // one tap per trip, NOT unrolled: unrolled, hipcc hoists all 9 * CIN * NCO scalar weight loads to the top and spills
  // hundreds of SGPRs through v_writelane/v_readlane
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, sx = tap - 3 * r;
    const bool ok = live && (unsigned)(y + r - 1) < (unsigned)H && (unsigned)(x + sx - 1) < (unsigned)W;
    // unconditional load from a mapped address + select (a branch around a load serialises the batch)
    const T* src = ok ? base + ((r - 1) * W + (sx - 1)) * LDI : in;
    float vv[4];
    if constexpr (sizeof(T) == 4) {
      const float4 v = *reinterpret_cast<const float4*>(src);
      vv[0] = ok ? v.x : 0.f, vv[1] = ok ? v.y : 0.f, vv[2] = ok ? v.z : 0.f, vv[3] = ok ? v.w : 0.f;
    } else {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
      for (int c = 0; c < 4; ++c) vv[c] = ok ? (float)v[c] : 0.f;
    }
    const float* wt = wf + tap * 4 * NCO;   // wave-uniform: scalar loads
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int co = 0; co < NCO; ++co) acc[co] = fmaf(vv[c], wt[c * NCO + co], acc[co]);
  }
  // Epilogue.  A lane holds the NCO channels of ONE pixel: stored directly, a 16-byte store instruction would touch 64
  // different lines.  Each wave transposes through its own LDS slab instead ([64 pixels][NCO + 4 floats], conflict-free
  // 16-byte writes and reads) so that NCO/4 consecutive lanes cover one pixel and a store instruction writes
  // 64 * 16 bytes of CONSECUTIVE pixels (the NHWC rows of a wave's 64 pixels are contiguous when ldout == NCO).
  __shared__ __attribute__((aligned(16))) float tr[4][64 * (NCO + 4)];
  float* slab = tr[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < NCO / 4; ++q) {
    float4 t;
    float* tp = &t.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = 4 * q + e;
      float vq = acc[co] * (scale ? scale[co] : 1.f) + (shift ? shift[co] : 0.f);
      tp[e] = relu ? fmaxf(vq, 0.f) : vq;
    }
    *reinterpret_cast<float4*>(slab + lane * (NCO + 4) + 4 * q) = t;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int CPL = sizeof(T) == 4 ? 4 : 8;   // channels per 16-byte store
  constexpr int LPP = NCO / CPL;          // lanes per pixel
  constexpr int PPI = 64 / LPP;           // pixels per store instruction
  const int64_t p0 = p - lane;            // first pixel of this wave (wave-uniform)
  const int qd = lane % LPP, pl = lane / LPP;
#pragma unroll
  for (int k = 0; k < LPP; ++k) {
    const int px = pl + k * PPI;          // pixel of the wave this lane stores in pass k
    const float4 t = *reinterpret_cast<const float4*>(slab + px * (NCO + 4) + CPL * qd);
    if constexpr (sizeof(T) == 4) {
      if (p0 + px < npix) *reinterpret_cast<float4*>(out + (p0 + px) * ldout + coff + 4 * qd) = t;
    } else {
      const float4 u = *reinterpret_cast<const float4*>(slab + px * (NCO + 4) + CPL * qd + 4);
      bf16x8 o;
      o[0] = (__bf16)t.x, o[1] = (__bf16)t.y, o[2] = (__bf16)t.z, o[3] = (__bf16)t.w;
      o[4] = (__bf16)u.x, o[5] = (__bf16)u.y, o[6] = (__bf16)u.z, o[7] = (__bf16)u.w;
      if (p0 + px < npix) *reinterpret_cast<bf16x8*>(out + (p0 + px) * ldout + coff + 8 * qd) = o;
    }
  }
}

// wf[tap][c][co] = w[co][c][tap] (OIHW), zero for c >= Cin
__global__ void pack_first_w_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin) {
  pack_first_w_body(w, wf, Cout, Cin, blockIdx.x, gridDim.x);
}

hipError_t launch_pack_first_w(const float* w, float* wf, int Cout, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(pack_first_w_kernel, dim3((9 * 4 * Cout + 255) / 256), dim3(256), 0, s, w, wf, Cout, Cin);
  return hipGetLastError();
}

bool first_conv_applicable(int dtype, int Cin, int Cp, int Cout, int ldout, int coff) {
  const int v = dtype == 0 ? 4 : 8;   // channels per 16 bytes
  return Cp == v && Cin >= 1 && Cin <= 4 && (Cout == 16 || Cout == 32 || Cout == 64) && (ldout % v) == 0 && (coff % v) == 0;
}

hipError_t launch_first_conv(int dtype, const void* in, const float* wf, const float* scale, const float* shift, void* out, int B, int H, int W,
                             int Cin, int Cout, int ldout, int coff, int relu, hipStream_t s) {
  const int64_t npix = (int64_t)B * H * W;
  const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
#define MGU_FC(T, NCO, CIN) hipLaunchKernelGGL((conv3x3_first_kernel<T, NCO, CIN>), grid, block, 0, s, (const T*)in, wf, scale, shift, (T*)out, npix, H, W, ldout, coff, relu)
#define MGU_FC_CIN(T, NCO)            \
  do {                                \
    if (Cin == 1) MGU_FC(T, NCO, 1);  \
    else if (Cin == 2) MGU_FC(T, NCO, 2); \
    else if (Cin == 3) MGU_FC(T, NCO, 3); \
    else MGU_FC(T, NCO, 4);           \
  } while (0)
#define MGU_FC_T(T)                   \
  do {                                \
    if (Cout == 16) MGU_FC_CIN(T, 16);      \
    else if (Cout == 32) MGU_FC_CIN(T, 32); \
    else if (Cout == 64) MGU_FC_CIN(T, 64); \
    else return hipErrorInvalidValue; \
  } while (0)
  if (dtype == 0) MGU_FC_T(float);
  else MGU_FC_T(__bf16);
#undef MGU_FC_T
#undef MGU_FC_CIN
#undef MGU_FC
  return hipGetLastError();
}

// ---- argmax over classes (first maximal index, like torch.argmax on distinct values) ---------------
__global__ void argmax_kernel(const float* __restrict__ logits, int64_t npix, int C, int64_t* __restrict__ pred) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = logits + i * C;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
      if (p[c] > best) {
        best = p[c];
        bi = c;
      }
    pred[i] = bi;
  }
}

hipError_t launch_argmax(const float* logits, int64_t npix, int C, int64_t* pred, hipStream_t s) {
  hipLaunchKernelGGL(argmax_kernel, dim3(nblocks(npix, 256)), dim3(256), 0, s, logits, npix, C, pred);
  return hipGetLastError();
}

}  // namespace mgu
