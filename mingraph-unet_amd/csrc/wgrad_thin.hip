// Weight gradients of the two THIN convolutions of the U-Net, whose GEMMs have no use for the matrix cores:
//   * the final 1x1 conv (unet_decoder.py:117,143):  dW[n][c] = sum_m dlogits[m][n] * feat[m][c],  N = 4 (padded classes)
//   * the first 3x3 conv (unet_encoder.py:7, Cin = 3 -> 4): dW[n][tap][ci] = sum_m dz[m][n] * x[pixel(m) + tap][ci], K = 36
// Both are one streaming pass over a 134 MB activation with ~1 FLOP per byte: HBM-bound work.  On the generic MFMA tile
// kernel (wgrad_f32.hip) they took 107 / 112 us each (K = 36 or N = 4 fills a fraction of a 32-wide tile and the
// split-M partial sums go through float atomics); here every thread keeps its outer-product slice in registers, a
// workgroup folds its pixel lanes through LDS once at the end and writes ONE partial panel (plain stores), in the
// same [group][n][Kp] format unpack_conv_grad_kernel already sums.
#include "common.h"

namespace mgu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- 1x1, N = 4:  thread = (pixel lane, channel quad) ------------------------------------------------------------
template <int QC>   // channel quads per pixel (Cp / 4): 8 or 16
__global__ __launch_bounds__(256) void wgrad_head_kernel(const float* __restrict__ z, int ldz, int zoff, const float* __restrict__ in,
                                                         int ldin, int inoff, int M, int Kp, float* __restrict__ dw, int rows_per_block) {
  constexpr int PL = 256 / QC;
  __shared__ float red[PL][QC][17];
  const int tid = threadIdx.x, cq = tid % QC, pl = tid / QC;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* zp = z + zoff;
  const float* xp = in + inoff + cq * 4;
  int m = m0 + pl;
  for (; m + 3 * PL < m1; m += 4 * PL) {   // four pixels in flight per thread
    f32x4 zv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      zv[u] = *reinterpret_cast<const f32x4*>(zp + (size_t)(m + u * PL) * ldz);
      xv[u] = *reinterpret_cast<const f32x4*>(xp + (size_t)(m + u * PL) * ldin);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] += zv[u][n] * xv[u];
  }
  for (; m < m1; m += PL) {
    const f32x4 zv = *reinterpret_cast<const f32x4*>(zp + (size_t)m * ldz);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + (size_t)m * ldin);
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] += zv[n] * xv;
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pl][cq][n * 4 + e] = acc[n][e];
  __syncthreads();
  if (tid < QC * 16) {
    const int q = tid >> 4, e = tid & 15;
    float s = 0.f;
#pragma unroll 8
    for (int p = 0; p < PL; ++p) s += red[p][q][e];
    dw[(size_t)blockIdx.x * 4 * Kp + (e >> 2) * Kp + q * 4 + (e & 3)] = s;
  }
}

// ---- 3x3 pad 1, Cp = 4, N = 32:  thread = (pixel lane, output-channel quad) ------------------------------------------
__global__ __launch_bounds__(256) void wgrad_first_kernel(const float* __restrict__ z, int ldz, int zoff, const float* __restrict__ in,
                                                          int ldin, int inoff, int M, int H, int W, int Kp, float* __restrict__ dw,
                                                          int rows_per_block) {
  __shared__ float red[32][8][17];
  const int tid = threadIdx.x, nq = tid & 7, pl = tid >> 3;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  f32x4 acc[9][4];   // [tap][ci] over this thread's 4 output channels
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) acc[t][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* zp = z + zoff + nq * 4;
  const float* xp = in + inoff;
  // software pipeline: the ten loads of the next pixel are in flight while the 144 FMAs of the current one issue
  f32x4 zv, xv[9];
  auto load_px = [&](int m, f32x4& zo, f32x4 (&xo)[9]) {
    const int mm = min(m, m1 - 1);             // past the end: re-read the last pixel (weight 0 below)
    const int x = mm % W, r = mm / W, y = r % H;
    zo = *reinterpret_cast<const f32x4*>(zp + (size_t)mm * ldz);
    if (m >= m1) zo = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {   // unconditional loads from a clamped address, zeroed by select (no branch around a load)
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const bool ok = (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (size_t)(ok ? mm + dy * W + dx : mm) * ldin);
      xo[t] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (m0 + pl < m1) {
    load_px(m0 + pl, zv, xv);
    for (int m = m0 + pl; m < m1; m += 32) {
      f32x4 zn, xn[9];
      load_px(m + 32, zn, xn);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) acc[t][ci] += xv[t][ci] * zv;
      zv = zn;
#pragma unroll
      for (int t = 0; t < 9; ++t) xv[t] = xn[t];
    }
  }
  // fold the 32 pixel lanes, one tap (4 ci x 4 n = 16 values per thread) per round
  float* const panel = dw + (size_t)blockIdx.x * 32 * Kp;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[pl][nq][ci * 4 + e] = acc[t][ci][e];
    __syncthreads();
    if (tid < 128) {
      const int q = tid >> 4, v = tid & 15;   // output-channel quad, (ci, e)
      float s = 0.f;
#pragma unroll 8
      for (int p = 0; p < 32; ++p) s += red[p][q][v];
      panel[(q * 4 + (v & 3)) * Kp + t * 4 + (v >> 2)] = s;
    }
    __syncthreads();
  }
}


bool wgrad_thin_applicable(const WgradDesc& d) {
  if (!tun(d).wgrad_thin || d.M < 4096 || (d.ldz & 3) || (d.zoff & 3) || (d.ldin & 3) || (d.inoff & 3)) return false;
  if (d.KS == 1 && d.N == 4 && (d.Cp == 32 || d.Cp == 64) && d.K == d.Cp) return true;
  if (d.KS == 3 && d.N == 32 && d.Cp == 4 && d.K == 36 && d.M == (d.M / (d.H * d.W)) * d.H * d.W) return true;
  return false;
}

hipError_t launch_wgrad_thin(WgradDesc& d, hipStream_t s) {
  int groups = 512;   // two workgroups per CU, each one contiguous slice of the pixels
  const int rows = (d.M + groups - 1) / groups;
  groups = (d.M + rows - 1) / rows;
  if ((size_t)groups * d.N * d.Kp > d.dw_capacity) return hipErrorInvalidValue;
  d.groups = groups;
  d.rows_per_split = rows;
  if (d.KS == 1) {
    if (d.Cp == 32)
      hipLaunchKernelGGL((wgrad_head_kernel<8>), dim3(groups), dim3(256), 0, s, d.z, d.ldz, d.zoff, d.in, d.ldin, d.inoff, d.M, d.Kp, d.dw, rows);
    else
      hipLaunchKernelGGL((wgrad_head_kernel<16>), dim3(groups), dim3(256), 0, s, d.z, d.ldz, d.zoff, d.in, d.ldin, d.inoff, d.M, d.Kp, d.dw, rows);
  } else {
    hipLaunchKernelGGL(wgrad_first_kernel, dim3(groups), dim3(256), 0, s, d.z, d.ldz, d.zoff, d.in, d.ldin, d.inoff, d.M, d.H, d.W, d.Kp, d.dw,
                       rows);
  }
  return hipGetLastError();
}

}  // namespace mgu
