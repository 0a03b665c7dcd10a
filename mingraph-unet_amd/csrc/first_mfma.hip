// First convolution of the U-Net (Conv2d(<= 3 -> 32, k3, p1) + folded bias / BatchNorm + ReLU, model/unet/unet_encoder.py:16-18 of the
// first ConvBlock) on the bf16 matrix cores.
//
// The VALU kernel it replaces on these shapes (conv3x3_first_kernel, elementwise.hip) spends 27 x 32 multiply-adds per pixel on the
// vector unit: 46 us of pure issue at 8 x 512^2, for a layer whose 0.3 GB of traffic take ~50 us (fp32) / ~30 us (bf16 storage).
// Here the layer is a GEMM with K = 27 padded to 32: M = pixels, N = 32 output channels, two k steps of v_mfma_f32_32x32x16_bf16.
//   * fp32 storage: both operands split exactly into three bf16 pieces (x3.h), six piece products per k step -- an fp32 GEMM in
//     accuracy, like the Winograd layers; bf16 storage: the activations ARE bf16 (one piece), the weights keep their three pieces
//     (three products), as accurate as the fp32-weight VALU kernel it replaces.
//   * a workgroup owns a 16 x 16 pixel patch: the 18 x 18 halo of packed pixels (one 16-byte load each: NHWC4 fp32 or NHWC8 bf16,
//     pack_input_kernel) goes to LDS once; a lane (pixel r = lane & 31 of a 2 x 16 pixel m tile, half h = lane >> 5) reads FIVE
//     16-byte pixels -- taps 0..4 (h = 0) or 5..8 (h = 1) -- which fill its 16 k slots: slot j of half h = (tap 5h + j / 3,
//     channel j % 3), slots past the 27 real values are zero; the weights are packed in the same slot order (pack_first_mfma_body);
//   * no barrier after the staging one, many small workgroups per CU cover each other's latencies; the epilogue stores 128 (64) bytes
//     of one pixel per half wave and instruction through a buffer descriptor (the halo kernel's form).
#include <algorithm>

#include "common.h"
#include "pack_small.h"
#include "x3.h"

namespace mgu {

// DIRECT: the halo is read from the caller's fp32 image through its element strides (any NCHW / NHWC view) and packed on the way to
// LDS -- pack_input_kernel's pass over the image (and the packed copy) disappears from the forward; bf16 storage rounds to nearest
// even exactly as that kernel does.
struct FirstSrc {
  const float* x;
  int64_t sn, sc, sh, sw;
  int cin;
};
template <typename T, bool DIRECT>
__global__ __launch_bounds__(256, 4) void conv3x3_first_mfma_kernel(const T* __restrict__ in, const FirstSrc src, const unsigned* __restrict__ wfm,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 T* __restrict__ out, int H, int W, int ldout, int coff, int relu,
                                                                 int tiles_x, int tiles_y, int nimg) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int LDI = F32 ? 4 : 8;        // elements of a packed input pixel (16 bytes)
  constexpr int HW = 18;
  __shared__ __attribute__((aligned(16))) f32x4 hs[HW * HW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = tiles_x * tiles_y * nimg;
  // the halo pixels of a patch this thread stages (two of the 324), as registers: requested for the NEXT patch of the workgroup's walk
  // while this one computes (a persistent workgroup: the first version -- one patch per workgroup, the other workgroups of the CU
  // covering the load -- took 78 us at 8 x 512^2, this one 66; a double-buffered halo with one barrier per patch needs more than 128
  // registers: three workgroups per CU, 76 us)
  f32x4 hv[2];
  auto load_patch = [&](const int p) {
    const int tx = p % tiles_x, ty = (p / tiles_x) % tiles_y, img = p / (tiles_x * tiles_y);
    const int y0 = ty * 16, x0 = tx * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hp = tid + 256 * i;
      const int hy = hp / HW, hx = hp - hy * HW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = hp < HW * HW && y >= 0 && y < H && x >= 0 && x < W;
      // unconditional load(s) from a mapped address + select (a branch around a load serialises the batch)
      f32x4 v;
      if constexpr (DIRECT) {
        const float* ps = src.x + (ok ? (int64_t)img * src.sn + (int64_t)y * src.sh + (int64_t)x * src.sw : 0);
        float ch[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ch[k] = ps[k < src.cin ? (ok ? k * src.sc : 0) : 0];
#pragma unroll
        for (int k = 0; k < 3; ++k) ch[k] = k < src.cin ? ch[k] : 0.f;
        if constexpr (F32) {
          v = f32x4{ch[0], ch[1], ch[2], 0.f};
        } else {
          const unsigned b0 = __builtin_bit_cast(unsigned short, (__bf16)ch[0]), b1 = __builtin_bit_cast(unsigned short, (__bf16)ch[1]),
                         b2 = __builtin_bit_cast(unsigned short, (__bf16)ch[2]);
          v = f32x4{__uint_as_float(b0 | (b1 << 16)), __uint_as_float(b2), 0.f, 0.f};
        }
      } else {
        v = *reinterpret_cast<const f32x4*>(ok ? in + ((size_t)img * H * W + (size_t)y * W + x) * LDI : in);
      }
      hv[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  load_patch(min((int)blockIdx.x, total - 1));
  // weight pieces of this lane's slots: [k step][piece][lane] x 16 bytes, resident in registers
  u32x4 bw[2][3];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) bw[s][pc] = *reinterpret_cast<const u32x4*>(wfm + ((size_t)(s * 3 + pc) * 64 + lane) * 4);
  const int lr = lane & 31, lh = lane >> 5;
  const float sc = scale ? scale[lr] : 1.f, sh = shift ? shift[lr] : 0.f;
  constexpr int SP = 36;                                  // slab pitch (floats)
  __shared__ __attribute__((aligned(16))) float slab_s[4][32 * SP];
  float* const slab = slab_s[wave];
  for (int p = blockIdx.x; p < total; p += gridDim.x) {
  const int tx = p % tiles_x, ty = (p / tiles_x) % tiles_y, img = p / (tiles_x * tiles_y);
  const int y0 = ty * 16, x0 = tx * 16;
  __syncthreads();   // every wave has read the previous patch's halo
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (tid + 256 * i < HW * HW) hs[tid + 256 * i] = hv[i];
  __syncthreads();
  load_patch(min(p + (int)gridDim.x, total - 1));   // unconditional (the last trip re-reads a patch): no branch around the loads
  f32x16 acc[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int prow = 2 * (2 * wave + mi) + (lr >> 4), pcol = lr & 15;   // this lane's pixel inside the patch
    float v[16];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int t = min(lh * 5 + q, 8);                 // tap of slot group q (half 1 has four: its fifth is masked below)
      const int dy = t / 3, dx = t - 3 * dy;
      const f32x4 px = hs[(prow + dy) * HW + pcol + dx];
      float c0, c1, c2;
      if constexpr (F32) {
        c0 = px[0], c1 = px[1], c2 = px[2];
      } else {   // 8 bf16: channels 0..2 in the first six bytes
        const unsigned w0 = __float_as_uint(px[0]), w1 = __float_as_uint(px[1]);
        c0 = __uint_as_float(w0 << 16), c1 = __uint_as_float(w0 & 0xffff0000u), c2 = __uint_as_float(w1 << 16);
      }
      const bool live = lh == 0 || q < 4;
      if (q < 5) {
        if (3 * q + 0 < 16) v[3 * q + 0] = live ? c0 : 0.f;
        if (3 * q + 1 < 16) v[3 * q + 1] = live ? c1 : 0.f;
        if (3 * q + 2 < 16) v[3 * q + 2] = live ? c2 : 0.f;
      }
    }
    v[15] = 0.f;
    f32x16 t16;
#pragma unroll
    for (int r = 0; r < 16; ++r) t16[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 pa[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (F32) {
          unsigned p0, p1, p2;
          split3_pack(v[8 * s + 2 * e], v[8 * s + 2 * e + 1], p0, p1, p2);
          pa[0][e] = p0, pa[1][e] = p1, pa[2][e] = p2;
        } else {   // the values are bf16 already: one piece
          pa[0][e] = __builtin_amdgcn_perm(__float_as_uint(v[8 * s + 2 * e + 1]), __float_as_uint(v[8 * s + 2 * e]), 0x07060302u);
        }
      }
      if constexpr (F32) {   // smallest products first (x3.h)
        t16 = mfma_bf16(pa[2], bw[s][0], t16);
        t16 = mfma_bf16(pa[0], bw[s][2], t16);
        t16 = mfma_bf16(pa[1], bw[s][1], t16);
        t16 = mfma_bf16(pa[1], bw[s][0], t16);
        t16 = mfma_bf16(pa[0], bw[s][1], t16);
        t16 = mfma_bf16(pa[0], bw[s][0], t16);
      } else {
        t16 = mfma_bf16(pa[0], bw[s][2], t16);
        t16 = mfma_bf16(pa[0], bw[s][1], t16);
        t16 = mfma_bf16(pa[0], bw[s][0], t16);
      }
    }
    acc[mi] = t16;
  }
  // ---- epilogue: accumulator register rr of m tile m = pixel (2m + (rr >> 3), 8 ((rr >> 2) & 1) + (rr & 3) + 4 lh), channel lr ----
  // A lane holds ONE channel of 16 pixels.  Interior patches transpose each m tile through the wave's own LDS slab ([32 pixels][32
  // channels + 4]) so that a lane stores 16 bytes and an instruction writes 1 KB of consecutive pixels of a patch row (dword stores
  // of two 128-byte lines per instruction: 82 us per launch at 8 x 512^2 against the 45 us of its bytes).
  T* const img_out = out + (size_t)img * H * W * ldout + coff;
  const bool interior = y0 + 16 <= H && x0 + 16 <= W;   // block-uniform
  const float relu_lo = relu ? 0.f : __builtin_nanf("");
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    float yv[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      float y = acc[mi][rr] * sc + sh;
      asm("v_max_f32_e32 %0, %1, %2" : "=v"(y) : "v"(relu_lo), "v"(y));   // one instruction; a quiet-NaN bound passes y through
      yv[rr] = y;
    }
    if (interior) {
      __builtin_amdgcn_wave_barrier();   // (the wave's LDS operations complete in order: the reads of the previous m tile are behind us)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) slab[(8 * (rr >> 2) + 4 * lh + (rr & 3)) * SP + lr] = yv[rr];   // pixel 16 (rr >> 3) + column
      __builtin_amdgcn_wave_barrier();
      constexpr int CPL = F32 ? 4 : 8;        // channels per 16-byte store
      constexpr int LPP = 32 / CPL;           // lanes per pixel
      constexpr int PPI = 64 / LPP;           // pixels per store instruction
      const int qd = lane % LPP, pl = lane / LPP;
#pragma unroll
      for (int k = 0; k < 32 / PPI; ++k) {
        const int px = pl + k * PPI;          // pixel of the m tile: patch row 2 (2 wave + mi) + (px >> 4), column px & 15
        const f32x4 t = *reinterpret_cast<const f32x4*>(slab + px * SP + CPL * qd);
        T* const o = img_out + ((size_t)(y0 + 2 * (2 * wave + mi) + (px >> 4)) * W + x0 + (px & 15)) * ldout + CPL * qd;
        if constexpr (F32) {
          __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(o));
        } else {
          const f32x4 u = *reinterpret_cast<const f32x4*>(slab + px * SP + CPL * qd + 4);
          bf16x8 ob;
          ob[0] = (__bf16)t[0], ob[1] = (__bf16)t[1], ob[2] = (__bf16)t[2], ob[3] = (__bf16)t[3];
          ob[4] = (__bf16)u[0], ob[5] = (__bf16)u[1], ob[6] = (__bf16)u[2], ob[7] = (__bf16)u[3];
          *reinterpret_cast<bf16x8*>(o) = ob;
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int pyc = 2 * (2 * wave + mi) + (rr >> 3), pxc = 8 * ((rr >> 2) & 1) + (rr & 3) + 4 * lh;
        if (y0 + pyc < H && x0 + pxc < W) img_out[((size_t)(y0 + pyc) * W + x0 + pxc) * ldout + lr] = (T)yv[rr];
      }
    }
  }
  }   // patch walk
}

__global__ void pack_first_mfma_kernel(const float* __restrict__ w, float* __restrict__ wfm, int Cout, int Cin) {
  pack_first_mfma_body(w, reinterpret_cast<uint16_t*>(wfm), Cout, Cin, blockIdx.x, gridDim.x);
}

size_t first_mfma_floats() { return 2 * 3 * 64 * 4; }   // [k step][piece][lane][4 dwords]

bool first_mfma_applicable(int dtype, int Cin, int Cp, int Cout, int ldout, int coff, int64_t H, int64_t W) {
  const int v = dtype == 0 ? 4 : 8;
  return Cp == v && Cin >= 1 && Cin <= 3 && Cout == 32 && (ldout % v) == 0 && (coff % v) == 0 &&
         H * W * ldout * (dtype == 0 ? 4 : 2) < 0x7ffffff0l;   // one image within the output descriptor's reach
}

hipError_t launch_pack_first_mfma(const float* w, float* wfm, int Cout, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(pack_first_mfma_kernel, dim3(1), dim3(256), 0, s, w, wfm, Cout, Cin);
  return hipGetLastError();
}

hipError_t launch_first_mfma(int dtype, const void* in, const float* wfm, const float* scale, const float* shift, void* out, int B, int H,
                             int W, int ldout, int coff, int relu, hipStream_t s) {
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  const dim3 grid((unsigned)std::min(tiles_x * tiles_y * B, 256 * 4)), block(256);   // four workgroups per CU are resident (launch bounds: 128 registers)
  const FirstSrc none{nullptr, 0, 0, 0, 0, 0};
  if (dtype == 0)
    hipLaunchKernelGGL((conv3x3_first_mfma_kernel<float, false>), grid, block, 0, s, (const float*)in, none, reinterpret_cast<const unsigned*>(wfm), scale,
                       shift, (float*)out, H, W, ldout, coff, relu, tiles_x, tiles_y, B);
  else
    hipLaunchKernelGGL((conv3x3_first_mfma_kernel<__bf16, false>), grid, block, 0, s, (const __bf16*)in, none, reinterpret_cast<const unsigned*>(wfm), scale,
                       shift, (__bf16*)out, H, W, ldout, coff, relu, tiles_x, tiles_y, B);
  return hipGetLastError();
}

// the same straight from the caller's image x (fp32, element strides sn / sc / sh / sw, cin <= 3 channels): no packed copy
hipError_t launch_first_mfma_direct(int dtype, const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int cin, const float* wfm,
                                    const float* scale, const float* shift, void* out, int B, int H, int W, int ldout, int coff, int relu,
                                    hipStream_t s) {
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  const dim3 grid((unsigned)std::min(tiles_x * tiles_y * B, 256 * 4)), block(256);
  const FirstSrc src{x, sn, sc, sh, sw, cin};
  if (dtype == 0)
    hipLaunchKernelGGL((conv3x3_first_mfma_kernel<float, true>), grid, block, 0, s, (const float*)nullptr, src, reinterpret_cast<const unsigned*>(wfm), scale,
                       shift, (float*)out, H, W, ldout, coff, relu, tiles_x, tiles_y, B);
  else
    hipLaunchKernelGGL((conv3x3_first_mfma_kernel<__bf16, true>), grid, block, 0, s, (const __bf16*)nullptr, src, reinterpret_cast<const unsigned*>(wfm), scale,
                       shift, (__bf16*)out, H, W, ldout, coff, relu, tiles_x, tiles_y, B);
  return hipGetLastError();
}

}  // namespace mgu
