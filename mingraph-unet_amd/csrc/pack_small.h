// Element loops of the small weight-packing kernels, callable with a VIRTUAL (block, grid) so that pack_wino_w_multi_kernel can run
// them as items of its one launch (a train step repacks every form after each optimizer step: the nine small launches of the U-Net
// cost more in launch gaps than in work).  The stand-alone kernels call the same bodies with (blockIdx.x, gridDim.x).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace mgu {

// wf[tap][c][co] = w[co][c][tap] (OIHW), zero for c >= Cin  (conv3x3_first_kernel, elementwise.hip)
__device__ __forceinline__ void pack_first_w_body(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin, unsigned vblock,
                                                  unsigned vgrid) {
  const int total = 9 * 4 * Cout;
  for (int i = (int)(vblock * blockDim.x + threadIdx.x); i < total; i += (int)(vgrid * blockDim.x)) {
    const int co = i % Cout, c = (i / Cout) % 4, tap = i / (4 * Cout);
    wf[i] = c < Cin ? w[((int64_t)co * Cin + c) * 9 + tap] : 0.f;
  }
}

__device__ __forceinline__ void bias_tile_body(const float* __restrict__ bias, float* __restrict__ shift, int C, int reps, unsigned vblock,
                                               unsigned vgrid) {
  for (int i = (int)(vblock * blockDim.x + threadIdx.x); i < C * reps; i += (int)(vgrid * blockDim.x)) shift[i] = bias[i % C];
}

// Three-piece fragment weights of the fp32 ConvTranspose (convt_x3.hip):
//   Wx[n / 128][k / 16][(n / 32) & 3][piece][lane = 32 * ((k / 8) & 1) + (n & 31)][k & 7]   (uint16 bf16 bit patterns)
// forward (dgrad = 0): n = (dy * 2 + dx) * Cout + co (the column order of the pixel-shuffle store), k = ci;
// data gradient (dgrad = 1): k = q * Cout + co (q = qy * 2 + qx), n = ci.  w is nn.ConvTranspose2d's (Cin, Cout, 2, 2).
__device__ __forceinline__ void pack_convt_x3_body(const float* __restrict__ w, uint16_t* __restrict__ Wx, int Cin, int Cout, int dgrad,
                                                   unsigned vblock, unsigned vgrid) {
  const int K = dgrad ? 4 * Cout : Cin;
  const int64_t total = (int64_t)Cin * Cout * 4;
  const int ksteps = K >> 4;
  for (int64_t idx = vblock * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)vgrid * blockDim.x) {
    const int k = (int)(idx % K), n = (int)(idx / K);
    const int q = dgrad ? k / Cout : n / Cout;
    const int co = dgrad ? k - q * Cout : n - q * Cout;
    const int ci = dgrad ? n : k;
    const float x = w[(((int64_t)ci * Cout + co) * 2 + (q >> 1)) * 2 + (q & 1)];
    const unsigned b0 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(b0);            // exact
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b1);           // exact; 8 significant bits are left
    const int lane = ((k >> 3) & 1) * 32 + (n & 31);
    uint16_t* dst = Wx + (((((int64_t)(n >> 7) * ksteps + (k >> 4)) * 4 + ((n >> 5) & 3)) * 3) * 64 + lane) * 8 + (k & 7);
    dst[0] = (uint16_t)(b0 >> 16);
    dst[512] = (uint16_t)(b1 >> 16);
    dst[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

// Three-piece weights of conv3x3_first_mfma_kernel (first_mfma.hip) in fragment order: [k step s][piece][lane][8] bf16 bit patterns;
// lane = 32 h + cout, element e of k step s = slot j = 8 s + e of half h = (tap 5 h + j / 3, channel j % 3); slots without a value
// (j = 15, taps > 8, channels >= Cin) are zero.  w is OIHW (32, Cin, 3, 3).
__device__ __forceinline__ void pack_first_mfma_body(const float* __restrict__ w, uint16_t* __restrict__ wfm, int Cout, int Cin, unsigned vblock,
                                                     unsigned vgrid) {
  for (int i = (int)(vblock * blockDim.x + threadIdx.x); i < 2 * 64 * 8; i += (int)(vgrid * blockDim.x)) {
    const int e = i & 7, lane = (i >> 3) & 63, s = i >> 9;
    const int co = lane & 31, h = lane >> 5, j = 8 * s + e;
    const int tap = 5 * h + j / 3, ch = j % 3;
    float x = 0.f;
    if (j < 15 && tap < 9 && ch < Cin && co < Cout) x = w[((int64_t)co * Cin + ch) * 9 + tap];
    const unsigned b0 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(b0);            // exact
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b1);           // exact; 8 significant bits are left
    uint16_t* dst = wfm + ((size_t)(s * 3) * 64 + lane) * 8 + e;
    dst[0] = (uint16_t)(b0 >> 16);
    dst[512] = (uint16_t)(b1 >> 16);
    dst[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

// data-gradient panel of a conv3x3 / 1x1 (train_kernels.hip): din = conv(dz, W') with W'[ci][(2-r,2-s), co] = W[co][ci][r][s]:
// panel [Cin][Kp], k = tap' * Cop + co  (Cop = Cout rounded up to 4, zero padded)
__device__ __forceinline__ void pack_dgrad_w_body(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int Cop, int KS, int Kp,
                                                  unsigned vblock, unsigned vgrid) {
  const int64_t total = (int64_t)Cin * Kp;
  for (int64_t i = vblock * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)vgrid * blockDim.x) {
    const int ci = (int)(i / Kp), k = (int)(i - (int64_t)ci * Kp);
    const int tap = k / Cop, co = k - tap * Cop;
    float v = 0.f;
    if (tap < KS * KS && co < Cout) v = w[((int64_t)co * Cin + ci) * KS * KS + (KS * KS - 1 - tap)];
    wp[i] = v;
  }
}

}  // namespace mgu
