// Winograd F(3x3, 2x2) weight gradient of the 3x3 / pad 1 convolutions on the fp32 matrix cores of gfx950.
//
// dW[co][ci][u][v] = sum over images and pixels of dz[co][y][x] * in[ci][y+u-1][x+v-1]  (the backward of
// model/unet/unet_encoder.py:16,20 through scripts/train_segmentation.py:130).  Per 2x2 tile of dz and its 4x4 input
// tile d this is a 3x3 correlation with a 2x2 "filter", which the Toom-Cook points (0, 1, -1, inf) turn into
//     dW += A^T [ (G dz G^T) .* (B^T d B) ] A,   G = [1 0; 1/2 1/2; 1/2 -1/2; 0 -1],   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]
// with the SAME input transform B^T d B as the forward Winograd kernel: 16 products per tile and (co, ci) pair instead of
// 36.  The 16 element-wise products, summed over tiles, are 16 GEMMs  M[ij][co][ci] = sum_tile S[ij][tile][co] V[ij][tile][ci]
// whose reduction dimension is the TILE index: v_mfma_f32_32x32x2_f32 with co as rows, ci as columns, two tiles per
// instruction.  The inverse transform A^T M A happens ONCE per workgroup, after its whole pixel range.
//
// MI355X mapping (same skeleton as wino_f32.hip)
//   * a workgroup of 4*CI_T wavefronts owns 32*CO_T output x 32*CI_T input channels (CO_T, CI_T in {1, 2}: 64 x 64 for
//     the bulk of the network, the narrower tiles for the 32-channel full-resolution layers) and a range of 8 x 16 pixel
//     patches (32 tiles = 16 MFMA k steps each); wavefront (i, g) owns transform row i and input-channel tile g, for
//     ALL of the workgroup's output-channel tiles;
//   * the raw input halo (10 x 18 px) and the dz patch (8 x 16 px) are staged once per patch in LDS;
//     neither S nor V is stored: a lane reads 2 x 2 dz values per output tile and 2 x 4 input values (4-byte reads; lanes
//     run along the channels, so a pixel pitch of 32*T + 16 floats puts the two tiles of a k step in different bank halves),
//     combines them with its row's coefficients and issues 8 MFMAs (4 components x 2 output tiles);
//   * accumulators: 4 components x 2 output tiles x 16 = 128 VGPRs, kept across all patches of the workgroup;
//   * epilogue: column part of A^T M A in registers, the four row waves meet through LDS, and the nine 3x3 taps are
//     written as plain 128-byte rows into a PRIVATE partial panel [co][tap * Cp + ci] of this patch group -- the format
//     of wgrad3x3_halo_f32_kernel, summed in a fixed order by unpack_conv_grad (bitwise reproducible, no atomics).
#include <type_traits>

#include "common.h"
#include "x3.h"

namespace mgu {


__device__ __forceinline__ void ww_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- inverse transform dW = A^T M A, once per workgroup: every wave is past its last LDS operand read when it gets here ----------
template <int CO_T, int CI_T>
__device__ __forceinline__ void ww_epilogue(const WgradDesc& d, float* smem, f32x16 (&acc)[4][CO_T], const int wi, const int wg, const int lane,
                                            const int lr, const int lh, const int cob, const int cib) {
  // column part in registers (q = 0..2 over j), row part through LDS (p = 0..2 over i):
  //   [q0 q1 q2] = [M0+M1+M2, M1-M2, M1+M2+M3];   [p0 p1 p2] = [Z0+Z1+Z2, Z1-Z2, Z1+Z2+Z3]
  float* Zx = smem;   // exchange: [4 rows i][CI_T g][4 register quads][64 lanes][4]   (<= 32 KB; the staging buffers are free)
  float* const part = d.dw + (size_t)blockIdx.x * d.N * d.Kp;
  const int co0 = cob * (32 * CO_T), ci0 = cib * (32 * CI_T) + wg * 32;
#pragma unroll
  for (int ct = 0; ct < CO_T; ++ct) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      ww_barrier();   // previous pass consumed (first pass: every wave is past its last LDS operand read)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rq + e;
          z[e] = q == 0 ? acc[0][ct][r] + acc[1][ct][r] + acc[2][ct][r]
               : q == 1 ? acc[1][ct][r] - acc[2][ct][r]
                        : acc[1][ct][r] + acc[2][ct][r] + acc[3][ct][r];
        }
        *reinterpret_cast<f32x4*>(Zx + ((((wi * CI_T + wg) * 4 + rq) * 64) + lane) * 4) = z;
      }
      ww_barrier();
      // wave i finishes accumulator registers 4i .. 4i+3: rows (co) 8i + 4*lh + (0..3) of output tile ct, column ci = lr
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(Zx + ((((0 * CI_T + wg) * 4 + wi) * 64) + lane) * 4);
      const f32x4 z1 = *reinterpret_cast<const f32x4*>(Zx + ((((1 * CI_T + wg) * 4 + wi) * 64) + lane) * 4);
      const f32x4 z2 = *reinterpret_cast<const f32x4*>(Zx + ((((2 * CI_T + wg) * 4 + wi) * 64) + lane) * 4);
      const f32x4 z3 = *reinterpret_cast<const f32x4*>(Zx + ((((3 * CI_T + wg) * 4 + wi) * 64) + lane) * 4);
      const f32x4 p0 = z0 + z1 + z2, p1 = z1 - z2, p2 = z1 + z2 + z3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + ct * 32 + 8 * wi + 4 * lh + e;
        float* row = part + (size_t)co * d.Kp + ci0 + lr;
        row[(0 * 3 + q) * d.Cp] = p0[e];
        row[(1 * 3 + q) * d.Cp] = p1[e];
        row[(2 * 3 + q) * d.Cp] = p2[e];
      }
    }
  }
}

// One k step (16 tiles: tile rows 2ks, 2ks+1) of the three-piece mode for wave (transform row i, input-channel tile g):
//   hread(sel, a, u, s4): raw input of halo row 2a + (sel ? rb : ra), column 4u + s4 of this lane's tiles, channel = lane
//   zread(ct, a, u, dy, dx): dz of pixel (2a + dy, 4u + dx), channel = lane of output tile ct
// (the lane half's two-pixel shift and the k-step's row offset are in the callers' base pointers).
template <int CO_T, class HRead, class ZRead>
__device__ __forceinline__ void ww_x3_kstep(f32x16 (&acc)[4][CO_T], const HRead& hread, const ZRead& zread, const float sgn, const float h0,
                                            const float h1) {
    // V values of this lane's 8 tiles, all four components of row i
    float vv[4][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = 4 * a + u;
        float rr[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) rr[s4] = x3_fma(sgn, hread(1, a, u, s4), hread(0, a, u, s4));
        vv[0][t] = x3_sub(rr[0], rr[2]), vv[1][t] = x3_add(rr[1], rr[2]), vv[2][t] = x3_sub(rr[2], rr[1]), vv[3][t] = x3_sub(rr[1], rr[3]);
      }
    // HALF of (G dz)[i][0], [i][1] per tile and output-channel tile: the 1/2 of the column step is folded in (exact)
    float cc[CO_T][2][8];
    auto load_c = [&](const int ct) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = 4 * a + u;
          cc[ct][0][t] = x3_fma(h1, zread(ct, a, u, 1, 0), h0 * zread(ct, a, u, 0, 0));
          cc[ct][1][t] = x3_fma(h1, zread(ct, a, u, 1, 1), h0 * zread(ct, a, u, 0, 1));
        }
    };
    load_c(0);
    u32x4 vb[4][3];   // V pieces [component j][piece], 8 packed tiles each: formed during the ct = 0 steps, reused by ct = 1
    u32x4 sa[2][3];   // S pieces of the step in flight / the next one
    // step = (output-channel tile ct, component j): its six MFMAs are issued between the split of the NEXT step's operands --
    // the wave issues in order, so six MFMAs back to back stall it for five MFMA times while the VALU idles (first version:
    // 28 % slower than the fp32 kernel it replaces)
    auto prep = [&](auto st_c) {
      constexpr int st = decltype(st_c)::value, ct = st >> 2, j = st & 3;
      if constexpr (ct == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned p0, p1, p2;
          split3_pack_s(vv[j][2 * e], vv[j][2 * e + 1], p0, p1, p2);
          vb[j][0][e] = p0, vb[j][1][e] = p1, vb[j][2][e] = p2;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const float p = cc[ct][0][2 * e + o], q = cc[ct][1][2 * e + o];
          x[o] = j == 0 ? x3_add(p, p) : j == 1 ? x3_add(p, q) : j == 2 ? x3_sub(p, q) : x3_sub(-q, q);
        }
        unsigned p0, p1, p2;
        split3_pack_s(x[0], x[1], p0, p1, p2);
        sa[st & 1][0][e] = p0, sa[st & 1][1][e] = p1, sa[st & 1][2][e] = p2;
      }
    };
    prep(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NS = 4 * CO_T;
    x3_static_for<0, NS>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value, ct = st >> 2, j = st & 3, sl = st & 1;
      f32x16 t = acc[j][ct];   // smallest products first
      t = mfma_bf16(sa[sl][2], vb[j][0], t);
      t = mfma_bf16(sa[sl][0], vb[j][2], t);
      t = mfma_bf16(sa[sl][1], vb[j][1], t);
      t = mfma_bf16(sa[sl][1], vb[j][0], t);
      t = mfma_bf16(sa[sl][0], vb[j][1], t);
      t = mfma_bf16(sa[sl][0], vb[j][0], t);
      acc[j][ct] = t;
      if constexpr (st + 1 < NS) prep(std::integral_constant<int, st + 1>{});
      if constexpr (CO_T == 2 && st == 1) load_c(1);
      constexpr int nvalu = (st + 1 < NS ? 52 : 0) + (st + 1 < 4 ? 44 : 0) + ((CO_T == 2 && st == 1) ? 32 : 0);
      constexpr int per = (nvalu + 5) / 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (CO_T == 2 && st == 1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if constexpr (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
}

template <int CO_T, int CI_T, bool X3>
__global__ __launch_bounds__(256 * CI_T, (X3 && CI_T == 1 && CO_T == 2) ? 2 : 1) void wino_wgrad_f32_kernel(const WgradDesc d, const int tiles_x, const int tiles_y,
                                                                    const int total_patches, const int patches_per_block,
                                                                    const int nci) {
  constexpr int NT = 256 * CI_T;               // threads
  constexpr int TH = 8, TW = 16, HWID = TW + 2, HPIX = (TH + 2) * HWID, ZPIX = TH * TW;
  constexpr int PH = 32 * CI_T + 16, PZ = 32 * CO_T + 16;   // floats per pixel in LDS: channels + 16 (bank-half offset between tiles)
  constexpr int QI = 8 * CI_T, QZ = 8 * CO_T;  // float4 per pixel: input halo / dz patch
  constexpr int HPP = NT / QI, ZPP = NT / QZ;  // pixels staged per pass
  constexpr int HR = (HPIX + HPP - 1) / HPP;   // float4 staged per thread: input halo
  constexpr int ZR = ZPIX / ZPP;               //                           dz patch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                // [HPIX][PH]
  float* Zs = smem + HPIX * PH;    // [ZPIX][PZ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, wg = wave >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  const int cib = blockIdx.y % nci;            // input-channel block  (32 * CI_T channels)
  const int cob = blockIdx.y / nci;            // output-channel block (32 * CO_T channels)
  const int p_begin = blockIdx.x * patches_per_block;
  const int npatch = min(patches_per_block, total_patches - p_begin);
  if (npatch <= 0) return;

  // transform row i:  B^T d: i=0: d0 - d2, i=1: d1 + d2, i=2: d2 - d1, i=3: d1 - d3   (rows ra + sgn * rb)
  //                   G dz : i=0: z0,      i=1: (z0 + z1)/2, i=2: (z0 - z1)/2, i=3: -z1  (a0 * z0 + a1 * z1)
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float sgn = wi == 1 ? 1.f : -1.f;
  const float a0 = wi == 0 ? 1.f : (wi == 3 ? 0.f : 0.5f);
  const float a1 = wi == 0 ? 0.f : (wi == 1 ? 0.5f : (wi == 2 ? -0.5f : -1.f));
  const float h0 = 0.5f * a0, h1 = 0.5f * a1;

  // ---- staging: thread -> (pixel, float4) of the halo and of the dz patch ----
  const int hq = tid % QI, hp0 = tid / QI;
  const int zq = tid % QZ, zp0 = tid / QZ;
  f32x4 hreg[HR], zreg[ZR];
  auto load_patch = [&](int p) {
    const int tx = p % tiles_x, ty = (p / tiles_x) % tiles_y, img = p / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const float* ibase = d.in + (size_t)img * d.H * d.W * d.ldin + d.inoff + cib * (32 * CI_T) + hq * 4;
    const float* zbase = d.z + (size_t)img * d.H * d.W * d.ldz + d.zoff + cob * (32 * CO_T) + zq * 4;
    // unconditional loads from a mapped address + select (a branch around a load serialises the batch)
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HPP * i;
      const int hy = hp / HWID, hx = hp - hy * HWID;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = hp < HPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? ibase + (size_t)(y * d.W + x) * d.ldin : d.in);
      hreg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
      const int zp = zp0 + ZPP * i;
      const int y = y0 + (zp >> 4), x = x0 + (zp & 15);
      const bool ok = y < d.H && x < d.W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? zbase + (size_t)(y * d.W + x) * d.ldz : d.z);
      zreg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < HR; ++i)
      if (hp0 + HPP * i < HPIX) *reinterpret_cast<f32x4*>(Hs + (hp0 + HPP * i) * PH + hq * 4) = hreg[i];
#pragma unroll
    for (int i = 0; i < ZR; ++i) *reinterpret_cast<f32x4*>(Zs + (zp0 + ZPP * i) * PZ + zq * 4) = zreg[i];
  };

  f32x16 acc[4][CO_T];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.f;

  // PF: the next patch is prefetched into registers while this one computes.  The three-piece 64 x 32 tile gives those 56 registers
  // to the operand pieces instead (256 incl. the accumulators = two workgroups per CU, which cover each other's load phase).
  constexpr bool PF = !(X3 && CI_T == 1 && CO_T == 2);
  if (PF) load_patch(p_begin);
  for (int pi = 0; pi < npatch; ++pi) {
    if (!PF) load_patch(p_begin + pi);
    ww_barrier();            // every wave is done with the previous patch
    store_patch();
    ww_barrier();            // patch visible
    // prefetch into registers while this patch computes -- UNCONDITIONAL (the last trip re-reads its own patch): a branch
    // around the loads makes hipcc wait for them at the join, i.e. before the MFMA loop instead of after it
    if (PF) load_patch(p_begin + min(pi + 1, npatch - 1));
    if constexpr (X3) {
      // Three-piece mode (the default): the SAME sums on the bf16 matrix pipe.  v_mfma_f32_32x32x16_bf16 reduces over 16 tiles per
      // instruction, a lane holding 8 of them per operand: lane half lh owns tiles (row 2ks + a, column 2u + lh), a = 0..1,
      // u = 0..3 -> k = 8 lh + 4a + u (the two halves stay two pixels apart, so the LDS reads keep their bank-half split).
      // S and V are formed as above, each value is split exactly into three bf16 pieces (x3.h) and a product is the six piece
      // products of weight >= 2^-16: 24 CO_T MFMAs of 32 cycles per 16 tiles instead of 32 CO_T of 64, paid for with ~700 VALU
      // operations -- the fp32 MFMA blocks the VALU while it runs, the bf16 one does not.
#pragma unroll 1
      for (int ks = 0; ks < 2; ++ks) {
        const float* const hks = Hs + (4 * ks * HWID + 2 * lh) * PH + wg * 32 + lr;   // tile rows 2ks, 2ks+1; this half's tile columns
        const float* const zks = Zs + (4 * ks * TW + 2 * lh) * PZ + lr;
        ww_x3_kstep<CO_T>(
            acc, [&](int sel, int a, int u, int s4) { return hks[((2 * a + (sel ? rb : ra)) * HWID + 4 * u + s4) * PH]; },
            [&](int ct, int a, int u, int dy, int dx) { return zks[((2 * a + dy) * TW + 4 * u + dx) * PZ + ct * 32]; }, sgn, h0, h1);
      }
    } else {
#pragma unroll 2
      for (int kk = 0; kk < 16; ++kk) {
        const int T = 2 * kk + lh;                 // this lane's tile of the k step (adjacent in x: bank halves differ)
        const int ty = T >> 3, tx = T & 7;
        // V[i][0..3] for this lane's input channel
        const float* hb = Hs + ((2 * ty) * HWID + 2 * tx) * PH + wg * 32 + lr;
        float rr[4];
  #pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) rr[s4] = hb[(ra * HWID + s4) * PH] + sgn * hb[(rb * HWID + s4) * PH];
        const float v0 = rr[0] - rr[2], v1 = rr[1] + rr[2], v2 = rr[2] - rr[1], v3 = rr[1] - rr[3];
  #pragma unroll
        for (int ct = 0; ct < CO_T; ++ct) {
          // S[i][0..3] for this lane's output channel of tile ct
          const float* zb = Zs + ((2 * ty) * TW + 2 * tx) * PZ + ct * 32 + lr;
          const float c0 = a0 * zb[0] + a1 * zb[TW * PZ];
          const float c1 = a0 * zb[PZ] + a1 * zb[TW * PZ + PZ];
          const float s0 = c0, s1 = 0.5f * (c0 + c1), s2 = 0.5f * (c0 - c1), s3 = -c1;
          acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, v0, acc[0][ct], 0, 0, 0);
          acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, v1, acc[1][ct], 0, 0, 0);
          acc[2][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s2, v2, acc[2][ct], 0, 0, 0);
          acc[3][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s3, v3, acc[3][ct], 0, 0, 0);
        }
      }
    }
  
    }
  ww_epilogue<CO_T, CI_T>(d, smem, acc, wi, wg, lane, lr, lh, cob, cib);
}

bool wino_wgrad_applicable(const WgradDesc& d) {
  return tun(d).wino_wgrad && d.KS == 3 && d.Cp % 32 == 0 && d.N % 32 == 0 && d.K == 9 * d.Cp && (d.ldin & 3) == 0 && (d.ldz & 3) == 0 &&
         (d.inoff & 3) == 0 && (d.zoff & 3) == 0 && (long)d.H * d.W * d.ldin < (1l << 31) && (long)d.H * d.W * d.ldz < (1l << 31) &&
         d.dw_capacity >= (size_t)d.N * d.Kp;
}

template <int CO_T, int CI_T, bool X3>
static hipError_t launch_ww_x(WgradDesc& d, hipStream_t s) {
  const int tiles_x = (d.W + 15) / 16, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B;
  const int nci = d.Cp / (32 * CI_T), nco = d.N / (32 * CO_T);
  // 1 (512 threads) or 2 (256 threads) workgroups per CU are resident; every patch group writes a private partial panel
  const int want = CI_T == 2 ? 256 : 512;
  int groups = (want + nci * nco - 1) / (nci * nco);
  const size_t cap_groups = d.dw_capacity / ((size_t)d.N * d.Kp);
  if ((size_t)groups > cap_groups) groups = (int)cap_groups;
  int ppb = (total + groups - 1) / groups;
  if (ppb < 1) ppb = 1;
  groups = (total + ppb - 1) / ppb;   // every group has >= 1 patch: every partial panel is fully written
  d.groups = groups;
  const size_t lds = (size_t)((8 + 2) * 18 * (32 * CI_T + 16) + 8 * 16 * (32 * CO_T + 16)) * sizeof(float);
  static bool attr_done[64] = {};
  hipError_t ae = ensure_dyn_lds(reinterpret_cast<const void*>(&wino_wgrad_f32_kernel<CO_T, CI_T, X3>), lds, attr_done);
  if (ae != hipSuccess) return ae;
  hipLaunchKernelGGL((wino_wgrad_f32_kernel<CO_T, CI_T, X3>), dim3(groups, nci * nco), dim3(256 * CI_T), lds, s, d, tiles_x, tiles_y, total,
                     ppb, nci);
  return hipGetLastError();
}

template <int CO_T, int CI_T>
static hipError_t launch_ww(WgradDesc& d, hipStream_t s) {
  if constexpr (CO_T == 2 && CI_T == 2) return launch_ww_x<2, 2, false>(d, s);   // three-piece mode never takes this tile (see below)
  else return tun(d).wgrad_x3 ? launch_ww_x<CO_T, CI_T, true>(d, s) : launch_ww_x<CO_T, CI_T, false>(d, s);
}

hipError_t launch_wino_wgrad_f32(WgradDesc& d, hipStream_t s) {
  const bool co2 = d.N % 64 == 0, ci2 = d.Cp % 64 == 0;
  // three-piece mode: the 64 x 64 tile needs 256 registers + spills (measured 28 % slower than its fp32-MFMA form), the 64 x 32
  // tile without register prefetch does not (254, two workgroups per CU cover each other's load phase).  A 64 x 64 tile with LDS-DMA
  // staging into a second LDS buffer (global_load_lds_dwordx4, XOR-swizzled dense images, counted vmcnt) was built and measured EQUAL
  // to it on every layer shape (88-92 us per 8.6 issued GFLOP either way): staging is not what bounds this kernel, VALU issue is
  if (tun(d).wgrad_x3 && co2) return launch_ww_x<2, 1, true>(d, s);
  if (co2 && ci2) return launch_ww<2, 2>(d, s);
  if (co2) return launch_ww<2, 1>(d, s);
  if (ci2) return launch_ww<1, 2>(d, s);
  return launch_ww<1, 1>(d, s);
}

}  // namespace mgu
