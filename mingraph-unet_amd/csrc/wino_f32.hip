// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 (CDNA4).
//
// Same operator as conv3x3_halo_kernel (igemm.hip) -- the 3x3 / pad 1 convolutions of ConvBlock
// (model/unet/unet_encoder.py:15-25) with the folded scale/shift/ReLU epilogue -- but with 2.25x fewer
// multiplications: every 2x2 output tile is computed from its 4x4 input tile as
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A                                  (Lavin & Gray, arXiv:1509.09308)
// so the 16 element-wise products, summed over input channels, are 16 independent GEMMs
//     M[ij][tile][cout] = sum_cin V[ij][tile][cin] * U[ij][cin][cout]
// which run on v_mfma_f32_32x32x2_f32 (exact fp32: the fp32 configuration of the reference is MFMA-bound, so
// removing multiplications is the only way past the 157 TFLOP/s matrix roofline).
//
// MI355X mapping
//   * a workgroup of 8 wavefronts owns an 8 x 32 pixel output patch (4 x 16 Winograd tiles = two 32-row MFMA
//     m tiles) and 32*NT output channels, and walks several patches (persistent);
//   * per 16 input channels the RAW 10 x 34 halo is staged once into LDS (double buffered) -- the transformed input V is never
//     written anywhere.  Wavefront (i, g) owns row i of the 4x4 transform: row i of B^T d is a +-1 combination of
//     two raw rows, so the wave reads 2 rows x 4 columns (8 ds_read_b128) per lane and k chunk, forms its 4
//     components V[i][0..3] with 8 vector adds, and issues 16 MFMAs on them.  That is 2 LDS reads per V value --
//     fewer than writing V to LDS and reading it back -- with no barrier between transform and MFMA;
//   * raw columns are stored split by parity ([row][x & 1][x >> 1][20 floats]): tile tx touches entries tx, tx+1 of
//     each parity plane, so the 16 lanes of a ds_read_b128 group are 80 bytes apart -> conflict free;
//   * U = G g G^T is precomputed (pack_wino_w_kernel) in MFMA-fragment order: the four components of a wave are one
//     contiguous 4 KB block per 8 input channels, loaded straight from L2 into VGPRs (coalesced 16-byte lanes).
//     Only one wave of the workgroup uses a given component, so staging U through LDS would buy nothing;
//   * inverse transform: each wave folds its own row (M[i][.] A) in registers, the four row waves meet through a
//     small LDS exchange, and lanes store one channel each (32 lanes = one 128-byte line of a pixel).
#include "common.h"
#include "pack_small.h"
#include "x3.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace mgu {

// a * b + c as ONE v_fma_f32 the backend cannot pair with a neighbour into v_pk_fma_f32
__device__ __forceinline__ float scalar_fma(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Workgroup barrier for LDS hand-offs ONLY.  __syncthreads() is a workgroup-scope fence + barrier, and the fence makes
// hipcc wait for vmcnt(0): every outstanding global load AND store (CDNA4 counts stores in vmcnt).  In this kernel that
// meant each epilogue barrier waited for the round trip of the output stores just issued, and each chunk barrier for the
// weight-fragment prefetch.  LDS operations of a wave complete in order, so lgkmcnt(0) before s_barrier is all a producer
// needs; the "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// max(x, 0) as ONE v_max_f32 (fmaxf() costs two: the backend first quiets a possible signalling NaN with v_max_f32 x, x, x).  Same
// result as fmaxf for every input, NaN included (the instruction returns the other operand).
__device__ __forceinline__ float relu_1op(const float x) {
  float r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Four LDS stores of one dword per lane at  M0 + OFF + 256 k + 4 * lane  (k = 0..3): ds_write_addtid_b32 takes no address VGPR
// and moves only the data dword to the LDS, half the store-path cycles of ds_write_b32 (MI355X_MICROARCH.md "LDS";
// tools/ubench/lds_addtid.hip: 2.9x on the exchange-write pattern of the Winograd epilogue, 8 waves per CU).  M0[15:0] is the
// base, so M0 + OFF reaches 131070 bytes: the callers choose OFF per region.  M0 is written inside the statement (one s_nop for
// the SALU-writes-M0 -> LDS add-TID hazard, which the hazard recognizer does not see inside inline asm) and nothing else in these
// kernels uses it (checked in the ISA: no LDS-DMA, no s_movrel, no GWS).
template <int OFF>
__device__ __forceinline__ void lds_store4_addtid(const unsigned m0v, const float a, const float b, const float c, const float e) {
  static_assert(OFF >= 0 && OFF + 768 <= 65535 && (OFF & 3) == 0, "ds_write_addtid_b32 offset is 16 bits");
  asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "ds_write_addtid_b32 %0 offset:%5\n\t"
               "ds_write_addtid_b32 %1 offset:%6\n\t"
               "ds_write_addtid_b32 %2 offset:%7\n\t"
               "ds_write_addtid_b32 %3 offset:%8"
               :
               : "v"(a), "v"(b), "v"(c), "v"(e), "s"(m0v), "n"(OFF), "n"(OFF + 256), "n"(OFF + 512), "n"(OFF + 768)
               : "memory");
}

// Operand precision of the 16 GEMMs (PREC):
//   0: v_mfma_f32_32x32x2_f32 on the fp32 operands;
//   1: every fp32 operand is split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits, by truncation:
//      a = a0 + a1 + a2) and the product is formed from the six piece products of weight >= 2^-16,
//        a b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a2 b0 + a1 b1),
//      on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped terms (a1 b2, a2 b1, a2 b2) are below
//      2^-23 |a b|, the size of one fp32 rounding, so this is an fp32 GEMM in accuracy; it costs 6 bf16 MFMA passes of
//      K = 16 (6 x 32 cycles) against 8 fp32 MFMAs of K = 2 (8 x 64 cycles) -- and, measured (tools/ubench/
//      mfma_valu.hip), the fp32 MFMA blocks the VALU while it runs whereas the bf16 MFMA does not, so the input
//      transform hides under the matrix pipe here.
// (Tuning::wino_prec; the packed U layout follows it, so launch_pack_wino_w takes the same value.)

// U[ntile][cin/8][i*4+j][lane (h = lane>>5, r = lane&31)][t]  =  (G g G^T)[i][j]  of  cout = 32*ntile + r,
// cin = 8*(cin/8) + 4*h + t.   dgrad = 1: the data-gradient conv, g'[u][v] = w[c][n][2-u][2-v] (roles swapped).
//
// prec = 1 (three bf16 pieces):  Ux[ntile][cin/16][i*4+j][piece][lane (h = lane>>5, r = lane&31)][e]  (uint16),
// cout = 32*ntile + r, cin = 16*(cin/16) + 8*h + e: the B fragment of v_mfma_f32_32x32x16_bf16, one 16-byte lane load.
__device__ __forceinline__ void pack_wino_w_body(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Cp, int Np,
                                                 int dgrad, int prec, unsigned vblock, unsigned vgrid) {
  if (prec == 1) {
    // Three-piece layout, store-coalesced: a thread owns output channel n and EIGHT consecutive input channels, i.e. one whole
    // 16-byte lane entry of every (component, piece) fragment; lane & 31 = n & 31 and lane >> 5 = the 8-channel half, so a wave's
    // store instruction writes one contiguous 1-KB fragment (a thread per (n, c) wrote 2-byte pieces 1 KB apart: 1.1 TB/s on the
    // 164 MB a train step re-packs).
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int nC = Cp >> 4;
    const int64_t total8 = (int64_t)Np * (Cp >> 3);
    for (int64_t idx = vblock * (int64_t)blockDim.x + threadIdx.x; idx < total8; idx += (int64_t)vgrid * blockDim.x) {
      const int lane = (int)(idx & 63);
      const int64_t grp = idx >> 6;
      const int c16 = (int)(grp % nC), ntile = (int)(grp / nC);
      const int n = ntile * 32 + (lane & 31), c0 = c16 * 16 + (lane >> 5) * 8;
      float g[8][3][3];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            const int c = c0 + e;
            float x = 0.f;
            if (n < Cout && c < Cin)
              x = dgrad ? w[(((int64_t)c * Cout + n) * 3 + (2 - u)) * 3 + (2 - v)] : w[(((int64_t)n * Cin + c) * 3 + u) * 3 + v];
            g[e][u][v] = x;
          }
      u32x4_t* dst = reinterpret_cast<u32x4_t*>(U) + ((int64_t)ntile * nC + c16) * 16 * 192 + lane;   // [comp][piece][64 lanes] x 16 B
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned short p0[8], p1[8], p2[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t3[3];   // row i of G g: t[i][v]
#pragma unroll
            for (int v = 0; v < 3; ++v)
              t3[v] = i == 0 ? g[e][0][v] : i == 3 ? g[e][2][v] : 0.5f * (g[e][0][v] + (i == 1 ? g[e][1][v] : -g[e][1][v]) + g[e][2][v]);
            const float uv = j == 0 ? t3[0] : j == 3 ? t3[2] : 0.5f * (t3[0] + (j == 1 ? t3[1] : -t3[1]) + t3[2]);
            const unsigned b0 = __float_as_uint(uv) & 0xffff0000u;
            const float r1 = uv - __uint_as_float(b0);               // exact
            const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(b1);               // exact; 8 significant bits are left
            p0[e] = (unsigned short)(b0 >> 16), p1[e] = (unsigned short)(b1 >> 16), p2[e] = (unsigned short)(__float_as_uint(r2) >> 16);
          }
          u32x4_t q0, q1, q2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            q0[e] = (unsigned)p0[2 * e] | ((unsigned)p0[2 * e + 1] << 16);
            q1[e] = (unsigned)p1[2 * e] | ((unsigned)p1[2 * e + 1] << 16);
            q2[e] = (unsigned)p2[2 * e] | ((unsigned)p2[2 * e + 1] << 16);
          }
          u32x4_t* q = dst + (i * 4 + j) * 192;
          q[0] = q0, q[64] = q1, q[128] = q2;
        }
    }
    return;
  }
  const int64_t total = (int64_t)Np * Cp;
  for (int64_t idx = vblock * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)vgrid * blockDim.x) {
    const int c = (int)(idx % Cp), n = (int)(idx / Cp);
    float g[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        float x = 0.f;
        if (n < Cout && c < Cin) {
          // forward: w is (Cout, Cin, 3, 3) and n = cout, c = cin.  dgrad: the layer's weight is (C_layer_out = Cin here,
          // C_layer_in = Cout here, 3, 3): output channel n of the dgrad conv is the layer's input channel.
          x = dgrad ? w[(((int64_t)c * Cout + n) * 3 + (2 - u)) * 3 + (2 - v)] : w[(((int64_t)n * Cin + c) * 3 + u) * 3 + v];
        }
        g[u][v] = x;
      }
    float t[4][3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      t[0][v] = g[0][v];
      t[1][v] = 0.5f * (g[0][v] + g[1][v] + g[2][v]);
      t[2][v] = 0.5f * (g[0][v] - g[1][v] + g[2][v]);
      t[3][v] = g[2][v];
    }
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i][0] = t[i][0];
      u[i][1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      u[i][2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      u[i][3] = t[i][2];
    }
    if (prec == 0) {
      float* dst = U + (((int64_t)(n >> 5) * (Cp >> 3) + (c >> 3)) * 16) * 256 + ((((c >> 2) & 1) * 32 + (n & 31)) * 4 + (c & 3));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(i * 4 + j) * 256] = u[i][j];
    } else {
      uint16_t* dst = reinterpret_cast<uint16_t*>(U) + (((int64_t)(n >> 5) * (Cp >> 4) + (c >> 4)) * 16) * (3 * 512) +
                      ((((c >> 3) & 1) * 32 + (n & 31)) * 8 + (c & 7));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned b0 = __float_as_uint(u[i][j]) & 0xffff0000u;
          const float r1 = u[i][j] - __uint_as_float(b0);            // exact
          const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
          const float r2 = r1 - __uint_as_float(b1);                 // exact; 8 significant bits are left
          uint16_t* q = dst + (i * 4 + j) * (3 * 512);
          q[0] = (uint16_t)(b0 >> 16);
          q[512] = (uint16_t)(b1 >> 16);
          q[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
        }
    }
  }
}

__global__ void pack_wino_w_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Cp, int Np,
                                   int dgrad, int prec) {
  pack_wino_w_body(w, U, Cout, Cin, Cp, Np, dgrad, prec, blockIdx.x, gridDim.x);
}

// Every Winograd weight set of a model in ONE launch (a train step re-packs 17 forward + 17 data-gradient sets after each
// optimizer step: 34 launches of ~5 us, most of them smaller than a launch gap).  The item table lives in device memory (the launcher uploads it when it changes: a
// 1.6 KB by-value kernel argument cost ~100 us of host time per launch).
__global__ void pack_wino_w_multi_kernel(const WinoPackBatch* __restrict__ bp) {
  const WinoPackBatch& b = *bp;
  int i = 0;
  while (i + 1 < b.n && blockIdx.x >= b.it[i + 1].blk0) ++i;   // <= 40 items: a linear scan of the block prefix
  const WinoPackItem& t = b.it[i];
  const unsigned vgrid = (i + 1 < b.n ? b.it[i + 1].blk0 : b.total_blocks) - t.blk0, vblock = blockIdx.x - t.blk0;
  switch (t.kind) {   // block-uniform
    case PACK_WINO: pack_wino_w_body(t.w, t.U, t.Cout, t.Cin, t.Cp, t.Np, t.dgrad, b.prec, vblock, vgrid); break;
    case PACK_FIRST_W: pack_first_w_body(t.w, t.U, t.Cout, t.Cin, vblock, vgrid); break;
    case PACK_CONVT_X3: pack_convt_x3_body(t.w, reinterpret_cast<uint16_t*>(t.U), t.Cin, t.Cout, t.dgrad, vblock, vgrid); break;
    case PACK_BIAS_TILE: bias_tile_body(t.w, t.U, t.Cout, t.Cin, vblock, vgrid); break;
    case PACK_DGRAD_W: pack_dgrad_w_body(t.w, t.U, t.Cout, t.Cin, t.Cp, t.dgrad, t.Np, vblock, vgrid); break;
    case PACK_FIRST_MFMA: pack_first_mfma_body(t.w, reinterpret_cast<uint16_t*>(t.U), t.Cout, t.Cin, vblock, vgrid); break;
    default: break;
  }
}

// n tiles padded to pairs; 6 bytes per value in the three-piece layout (sized for either)
size_t wino_u_floats(int Cout, int Cp) { return (size_t)((Cout + 63) / 64 * 64) * Cp * 24; }

// fills the launcher-owned fields of a batch (Np, block prefix); false if an item is not packable
bool wino_pack_batch_prepare(WinoPackBatch& b) {
  unsigned blk = 0;
  for (int i = 0; i < b.n; ++i) {
    WinoPackItem& t = b.it[i];
    t.blk0 = blk;
    int64_t work;   // threads' worth of elements
    switch (t.kind) {
      case PACK_WINO:
        if (t.Cp & (b.prec ? 15 : 7)) return false;
        t.Np = (t.Cout + 63) / 64 * 64;
        work = (int64_t)t.Np * t.Cp / (b.prec ? 8 : 1);
        break;
      case PACK_FIRST_W: work = 9 * 4 * (int64_t)t.Cout; break;
      case PACK_CONVT_X3:
        if ((t.Cin & 31) || (t.Cout & 31)) return false;
        work = (int64_t)t.Cin * t.Cout * 4;
        break;
      case PACK_BIAS_TILE: work = (int64_t)t.Cout * t.Cin; break;
      case PACK_DGRAD_W: work = (int64_t)t.Cin * t.Np; break;
      case PACK_FIRST_MFMA: work = 2 * 64 * 8; break;
      default: return false;
    }
    blk += (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (work + 255) / 256));
  }
  b.total_blocks = blk;
  return true;
}
hipError_t launch_pack_wino_w_multi(const WinoPackBatch* batch_dev, unsigned total_blocks, hipStream_t s) {
  if (total_blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_wino_w_multi_kernel, dim3(total_blocks), dim3(256), 0, s, batch_dev);
  return hipGetLastError();
}

hipError_t launch_pack_wino_w(const float* w, float* U, int Cout, int Cin, int Cp, int dgrad, int prec, hipStream_t s) {
  if (Cp & (prec ? 15 : 7)) return hipErrorInvalidValue;
  const int Np = (Cout + 63) / 64 * 64;
  int64_t blocks = ((int64_t)Np * Cp + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(pack_wino_w_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, U, Cout, Cin, Cp, Np, dgrad, prec);
  return hipGetLastError();
}

// Work split (MODE):
//   0: 8 wavefronts, 64 output channels; wave (i, g) owns transform row i, n tile g and both m tiles of the patch
//   1: 8 wavefronts, 32 output channels; wave (i, g) owns row i and m tile g
//   (a 4-wavefront split with two independent workgroups per CU was measured too: no faster, and it spills)
//
// Pipeline: the raw halo streams in 16-channel chunks through TWO LDS buffers.  At the top of chunk c (one barrier)
// every thread parks chunk c+1 (loaded one chunk ago into VGPRs) in the idle buffer and issues the loads of chunk
// c+2; the weight fragments of k group g+1 are issued before the MFMAs of group g.  Nothing is waited for in the step
// it was issued in.
template <int MODE, int PREC>
__global__ __launch_bounds__(512) void wino3x3_f32_kernel(const IgemmDesc d, const int tiles_x, const int tiles_y, const int total_patches,
                        const int patches_per_block, const int ngroups, const int nitems, const int per_xcd) {
  constexpr int NWAVES = 8;
  constexpr int MT = MODE == 1 ? 1 : 2;          // m tiles per wavefront
  constexpr int NTB = MODE == 0 ? 2 : 1;         // n tiles (32 output channels) per workgroup
  constexpr int NC = 32 * NTB;                   // output channels per workgroup
  constexpr int ZP = NC + 8;                     // exchange-buffer pitch of a tile (floats)
  constexpr int QPT = NC / 4;                    // channel quads per tile
  constexpr int UPT = 64 * QPT / 512;            // (tile, channel quad) units a thread finishes per pass
  constexpr int RH = 10, RW = 34, HPIX = RH * RW;   // raw halo of the 8 x 32 pixel patch
  constexpr int PLD = 20;                        // floats per raw pixel in LDS: 16 channels + 4 pad (80 bytes: 16 lanes
                                                 // of a ds_read_b128 group, one pixel apart, hit 16 distinct 4-bank slots)
  constexpr int HSTRIDE = NWAVES * 16;           // raw pixels staged per pass (4 threads x 16 bytes per pixel)
  constexpr int HR = (HPIX + HSTRIDE - 1) / HSTRIDE;
  constexpr int S1 = 17 * PLD, S2 = PLD, S3 = 17 * PLD + PLD;   // LDS offsets of tile columns 1..3 (parity planes)
  constexpr int RAWF = HR * HSTRIDE / RW * 34 * PLD + 34 * PLD;   // floats per raw buffer, padded: the staging pass of the
                                                 // threads past HPIX stores (never read) dummies instead of branching
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zx = smem + 2 * RAWF;      // [4 rows i][64 tiles][ZP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, wg = wave >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  const int tx = lr & 15, ty = lr >> 4;
  // XCD-aware work order: workgroup L runs on XCD L % 8 (round-robin dispatch); give each XCD a CONTIGUOUS range of
  // (n block, patch group) items in n-major order, so the workgroups that share an L2 stream the same U slice (large
  // layers: 1-4 MB per n block against a 4 MB L2) and neighbouring patches (shared halo rows).
  const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (item >= nitems) return;
  const int nblock = item / ngroups;
  const int p_begin = (item - nblock * ngroups) * patches_per_block;
  const int npatch = min(patches_per_block, total_patches - p_begin);
  if (npatch <= 0) return;

  // row i of B^T d:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float sgn = wi == 1 ? 1.f : -1.f;
  int offA[MT], offB[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int m_abs = MODE == 1 ? wg : mi;
    const int rowbase = 2 * (2 * m_abs + ty);
    offA[mi] = ((rowbase + ra) * 34 + tx) * PLD + lh * (PREC ? 8 : 4);
    offB[mi] = ((rowbase + rb) * 34 + tx) * PLD + lh * (PREC ? 8 : 4);
  }
  const int ncg = d.Cp >> 3;                       // 8-channel k groups
  const int nC = d.Cp >> 4;                        // 16-channel raw chunks
  const int ntg = nblock * NTB + (MODE == 0 ? wg : 0);
  const float* const up = d.wu + ((size_t)ntg * ncg * 16 + wi * 4) * 256 + lane * 4;
  // three-piece layout: 16-byte lane loads, [chunk][i*4+j][piece][lane]
  const u32x4* const upx = reinterpret_cast<const u32x4*>(d.wu) + ((size_t)ntg * nC * 16 + wi * 4) * 192 + lane;

  // ---- raw halo staging: thread -> (pixel hp0 + HSTRIDE i, 16-byte piece kq) ----
  const int kq = tid & 3, hp0 = tid >> 2;
  int hoff[HR];
  unsigned hmask = 0u, hmask_next = 0u;
  const float* load_base = d.in;
  auto setup_patch = [&](int p, int& img, int& y0, int& x0) {
    const int px = p % tiles_x;
    const int py = (p / tiles_x) % tiles_y;
    img = p / (tiles_x * tiles_y);
    y0 = py * 8;
    x0 = px * 32;
  };
  auto setup_load = [&](int p) {
    int img, y0, x0;
    setup_patch(p, img, y0, x0);
    load_base = d.in + (size_t)img * d.H * d.W * d.ldin + kq * 4;
    unsigned mk = 0u;
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HSTRIDE * i;
      const int r = hp / RW, cc = hp - r * RW;
      const int y = y0 - 1 + r, x = x0 - 1 + cc;
      const bool ok = hp < HPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
      hoff[i] = ok ? (y * d.W + x) * d.ldin : 0;   // unconditional loads from a mapped address; zeroed at the LDS store
      mk |= ok ? (1u << i) : 0u;
    }
    hmask_next = mk;
  };
  f32x4 hreg[HR];
  auto load_halo = [&](int c) {
    hmask = hmask_next;
#pragma unroll
    for (int i = 0; i < HR; ++i) hreg[i] = *reinterpret_cast<const f32x4*>(load_base + hoff[i] + c * 16);
  };
  auto store_halo = [&](float* Hs) {
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HSTRIDE * i;
      const int r = hp / RW, cc = hp - r * RW;
      *reinterpret_cast<f32x4*>(Hs + ((r * 2 + (cc & 1)) * 17 + (cc >> 1)) * PLD + kq * 4) =
          ((hmask >> i) & 1u) ? hreg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // the load stream runs two chunks ahead of the compute: (lp, lc) = patch / chunk of the NEXT load.
  // (the loads themselves are UNCONDITIONAL: a branch around a load makes hipcc drain vmcnt at the join, which would
  //  wait for the chunk just issued.  Past the end of the stream the last patch is simply re-read and never used.)
  int lp = 0, lc = 0;
  auto prep_next = [&]() {   // address setup of the next load (branchy: kept out of the chunk body)
    if (lc == 0 && lp < npatch) setup_load(p_begin + lp);
  };
  auto load_next = [&]() {   // issue the loads of the next chunk in stream order (straight-line code)
    load_halo(lc);
    lc = lc + 1 == nC ? 0 : lc + 1;
    lp += lc == 0 ? 1 : 0;
  };

  // finishing role of this thread in the epilogue: channel quad cq of the workgroup's NC channels (same for all its units)
  const int cq = tid % QPT;
  const int n0 = nblock * NC + cq * 4;
  // 16-byte stores need whole, aligned channel quads
  const bool fast_n = (d.N % NC == 0) && (d.ldout % 4 == 0) && (d.coff % 4 == 0) && (!d.pool || d.ldpool % 4 == 0);
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n0 + e < d.N) {
      if (d.scale) sc4[e] = d.scale[n0 + e];
      if (d.shift) sh4[e] = d.shift[n0 + e];
    }

  f32x4 bf[2][4];
  auto load_b = [&](int cg, int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[slot][j] = *reinterpret_cast<const f32x4*>(up + (size_t)cg * 4096 + j * 256);
  };
  u32x4 bx[4][3];   // PREC 1: the three pieces of this wave's four components, one 16-channel chunk
  auto load_bx = [&](int j, int chunk) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) bx[j][pc] = upx[(size_t)chunk * (16 * 192) + j * 192 + pc * 64];
  };

  f32x16 acc[4][MT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][mi][r] = 0.f;

  prep_next();
  load_next();              // chunk 0 of the first patch
  if constexpr (PREC == 0) {
    load_b(0, 0);
    load_b(1, 1);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) load_bx(j, 0);
  }
  store_halo(smem);
  prep_next();
  load_next();              // chunk 1 of the stream, parked at the top of chunk 0
  int cg = 0;               // k group whose fragments sit in bf[0] at the top of a chunk
  f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};   // training: sum z, sum z^2 of this thread's channel quad
  int buf = 0;              // raw buffer of the chunk being computed
  for (int pi = 0; pi < npatch; ++pi) {
    for (int c = 0; c < nC; ++c) {
      prep_next();
      lds_barrier();   // the chunk to compute is visible in buffer buf; every wave has left buffer buf ^ 1
      const float* Hs = smem + buf * RAWF;
      if constexpr (PREC == 0) {
        auto operands = [&](const int kg, const int mi, f32x4 (&v)[4]) {   // V[i][0..3] of this lane's tile, 4 channels
          const float* pa = Hs + offA[mi] + kg * 8;
          const float* pb = Hs + offB[mi] + kg * 8;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(pa) + sgn * *reinterpret_cast<const f32x4*>(pb);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(pa + S1) + sgn * *reinterpret_cast<const f32x4*>(pb + S1);
          const f32x4 r2 = *reinterpret_cast<const f32x4*>(pa + S2) + sgn * *reinterpret_cast<const f32x4*>(pb + S2);
          const f32x4 r3 = *reinterpret_cast<const f32x4*>(pa + S3) + sgn * *reinterpret_cast<const f32x4*>(pb + S3);
          v[0] = r0 - r2;
          v[1] = r1 + r2;
          v[2] = r2 - r1;
          v[3] = r1 - r3;
        };
        auto mfma16 = [&](const f32x4 (&v)[4], const int slot, const int mi) {
  #pragma unroll
          for (int j = 0; j < 4; ++j)
  #pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[j][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][t], bf[slot][j][t], acc[j][mi], 0, 0, 0);
        };
        // The staging of a chunk -- 3 LDS stores (chunk c+1), the next k group's 4 fragment loads, 3 halo loads (chunk
        // c+2) -- has no dependence on this chunk's MFMAs.  Left as a phase of its own it costs ~1500 cycles per chunk in
        // which NO wave of the workgroup feeds the matrix pipe (all eight are in the same phase, by the barrier); a wave
        // has ~60 free issue cycles behind every MFMA, so the staging instructions are threaded between the first m
        // tile's MFMAs instead (sched_group_barrier pattern below).  Issue order: fragment loads BEFORE halo loads (vmcnt
        // retires in order; the halo is the long-latency stream and must not sit in front of the kg = 1 fragment wait).
        // Fragment prefetch distance is ONE WHOLE CHUNK (slot 0 is reloaded at the start of the second half, slot 1 at
        // the end of the chunk): vmcnt counts stores too and retires in order, so a load issued after the epilogue's
        // output stores cannot be waited for before those stores have gone all the way to memory (~2-3 us).  With both
        // k groups of a patch's first chunk already in flight BEFORE the previous epilogue, nothing issued behind the
        // stores is needed for at least half a chunk.
        f32x4 va[4], vb[4];
        operands(0, 0, va);
        store_halo(smem + (buf ^ 1) * RAWF);
        load_next();
        mfma16(va, 0, 0);
        if (MT == 2) {
          operands(0, 1, vb);
          mfma16(vb, 0, 1);
        }
  #pragma unroll
        for (int i = 0; i < HR; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
  #pragma unroll
        for (int i = 0; i < HR; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if (MT == 2) {
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        int nx = cg + 2 >= ncg ? cg + 2 - ncg : cg + 2;
        cg = nx;
        operands(1, 0, va);
        load_b(nx, 0);   // first k group of the next chunk (wraps to the next patch's first)
        mfma16(va, 1, 0);
        if (MT == 2) {
          operands(1, 1, vb);
          mfma16(vb, 1, 1);
        }
        load_b(nx + 1 == ncg ? 0 : nx + 1, 1);   // second k group of the next chunk, issued as its slot drains
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
        }
        if (MT == 2) {
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (MT == 2) {
        // three-piece path: one K = 16 slab per chunk.  Lane (tile, h) forms V[i][0..3] for channels 8h .. 8h+7 (two
        // 4-channel halves), splits each component into packed bf16 pieces and issues the six piece products.
        // (Two m tiles per wave leave no registers for the software pipeline of the MT == 1 branch below -- it spills
        //  105 VGPRs -- so this one is left to the compiler's own schedule: measured 15-20 % faster than the fp32 MFMA
        //  path on the deep layers.)
        store_halo(smem + (buf ^ 1) * RAWF);
        load_next();
        const int cn = c + 1 == nC ? 0 : c + 1;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          f32x4 v[4][2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float* pa = Hs + offA[mi] + hf * 4;
            const float* pb = Hs + offB[mi] + hf * 4;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(pa) + sgn * *reinterpret_cast<const f32x4*>(pb);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(pa + S1) + sgn * *reinterpret_cast<const f32x4*>(pb + S1);
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(pa + S2) + sgn * *reinterpret_cast<const f32x4*>(pb + S2);
            const f32x4 r3 = *reinterpret_cast<const f32x4*>(pa + S3) + sgn * *reinterpret_cast<const f32x4*>(pb + S3);
            v[0][hf] = r0 - r2;
            v[1][hf] = r1 + r2;
            v[2][hf] = r2 - r1;
            v[3][hf] = r1 - r3;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x4 a0, a1, a2;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                unsigned p0, p1, p2;
                split3_pack(v[j][hf][2 * e], v[j][hf][2 * e + 1], p0, p1, p2);
                a0[hf * 2 + e] = p0, a1[hf * 2 + e] = p1, a2[hf * 2 + e] = p2;
              }
            f32x16 t = acc[j][mi];
            t = mfma_bf16(a2, bx[j][0], t);
            t = mfma_bf16(a0, bx[j][2], t);
            t = mfma_bf16(a1, bx[j][1], t);
            t = mfma_bf16(a1, bx[j][0], t);
            t = mfma_bf16(a0, bx[j][1], t);
            t = mfma_bf16(a0, bx[j][0], t);
            acc[j][mi] = t;
            if (mi == MT - 1) load_bx(j, cn);   // this component's pieces of the next chunk (or the next patch's first)
          }
        }
      } else {
        // three-piece path, one m tile per wave: ~250 VALU instructions against 24 MFMAs per chunk and wave, so the VALU
        // is the busier pipe; the code is software-pipelined in four steps (component j): the MFMAs of step s are
        // interleaved with the split of step s+1, which the bf16 MFMA lets the VALU do concurrently.
        const int cn = c + 1 == nC ? 0 : c + 1;
        f32x4 v[2][4][2];   // [m tile][j][half]
        auto transform = [&](const int mi, const int hf) {
          const float* pa = Hs + offA[mi] + hf * 4;
          const float* pb = Hs + offB[mi] + hf * 4;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(pa) + sgn * *reinterpret_cast<const f32x4*>(pb);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(pa + S1) + sgn * *reinterpret_cast<const f32x4*>(pb + S1);
          const f32x4 r2 = *reinterpret_cast<const f32x4*>(pa + S2) + sgn * *reinterpret_cast<const f32x4*>(pb + S2);
          const f32x4 r3 = *reinterpret_cast<const f32x4*>(pa + S3) + sgn * *reinterpret_cast<const f32x4*>(pb + S3);
          v[mi][0][hf] = r0 - r2;
          v[mi][1][hf] = r1 + r2;
          v[mi][2][hf] = r2 - r1;
          v[mi][3][hf] = r1 - r3;
        };
        u32x4 pc[2][3];     // [step parity][piece]
        auto split = [&](const int mi, const int j, const int slot) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              unsigned p0, p1, p2;
              split3_pack(v[mi][j][hf][2 * e], v[mi][j][hf][2 * e + 1], p0, p1, p2);
              pc[slot][0][hf * 2 + e] = p0, pc[slot][1][hf * 2 + e] = p1, pc[slot][2][hf * 2 + e] = p2;
            }
        };
        auto mfma6 = [&](const int mi, const int j, const int slot) {
          f32x16 t = acc[j][mi];
          t = mfma_bf16(pc[slot][2], bx[j][0], t);
          t = mfma_bf16(pc[slot][0], bx[j][2], t);
          t = mfma_bf16(pc[slot][1], bx[j][1], t);
          t = mfma_bf16(pc[slot][1], bx[j][0], t);
          t = mfma_bf16(pc[slot][0], bx[j][1], t);
          t = mfma_bf16(pc[slot][0], bx[j][0], t);
          acc[j][mi] = t;
        };
        transform(0, 0);
        transform(0, 1);
        split(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NS = 4 * MT;
        static_for<0, NS>([&](auto st_c) {
          constexpr int st = decltype(st_c)::value;
          constexpr int mi = st >> 2, j = st & 3;
          mfma6(mi, j, st & 1);
          if constexpr (st + 1 < NS) split((st + 1) >> 2, (st + 1) & 3, (st + 1) & 1);
          if constexpr (MT == 2 && st == 1) transform(1, 0);
          if constexpr (MT == 2 && st == 2) transform(1, 1);
          if constexpr (st == NS - 2) {          // staging of the next chunks rides in a late step
            store_halo(smem + (buf ^ 1) * RAWF);
            load_next();
          }
          if constexpr (mi == MT - 1 && j > 0) load_bx(j - 1, cn);   // pieces of the next chunk (or the next patch's first)
          constexpr int nvalu = (st + 1 < NS ? 44 : 0) + ((MT == 2 && (st == 1 || st == 2)) ? 32 : 0) + (st == NS - 2 ? 16 : 0);
          constexpr int per = (nvalu + 5) / 6;
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (MT == 2 && (st == 1 || st == 2)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if constexpr (per > 0) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
            if constexpr (st == NS - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if constexpr (st == NS - 2 || (mi == MT - 1 && j > 0)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        load_bx(3, cn);
      }
      buf ^= 1;
    }
    // ---- inverse transform + epilogue of patch pi --------------------------------------------------------
    // The column part (M[i][.] A) is folded in registers; the four row waves then meet in LDS.  The exchange buffer is
    // [row i][tile 0..63][channel] (channel fastest, pitch NC + 8: the two 32-lane halves of a store land in different
    // bank halves), so the FINISHING threads can each take one tile and FOUR consecutive channels: 16-byte loads of the
    // four rows' partial sums, and 16-byte global stores in which 16 (8) lanes cover one pixel's 256 (128) contiguous
    // bytes -- a quarter of the store instructions of a lane-per-channel epilogue, all full lines.  (Store instruction
    // issue, not bytes, was what made the epilogue cost as much as a whole 16-channel chunk.)
    int img, y0, x0;
    setup_patch(p_begin + pi, img, y0, x0);
    float* const img_out = d.out + (size_t)img * d.H * d.W * d.ldout + d.coff;
    const unsigned sW = (unsigned)(d.W * d.ldout);
    const bool interior = (y0 + 8 <= d.H) && (x0 + 32 <= d.W) && fast_n;   // block-uniform
    float* pool_out = nullptr;   // fused MaxPool2d(2): a Winograd tile IS a pooling window
    if (d.pool) pool_out = d.pool + (size_t)img * (d.H >> 1) * (d.W >> 1) * d.ldpool;
    f32x4 pmax[UPT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // column part: Z[i][q] = sum_j M[i][j] A[j][q]   (A^T = [1 1 1 0; 0 1 -1 -1])
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int m_abs = MODE == 1 ? wg : mi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = q == 0 ? (acc[0][mi][r] + acc[1][mi][r] + acc[2][mi][r]) : (acc[1][mi][r] - acc[2][mi][r] - acc[3][mi][r]);
          const int T = m_abs * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          Zx[(wi * 64 + T) * ZP + (MODE == 0 ? wg * 32 : 0) + lr] = z;
        }
      }
      lds_barrier();
      // row part + epilogue: unit u = (tile, channel quad); y(2tr, .) = Z0 + Z1 + Z2, y(2tr+1, .) = Z1 - Z2 - Z3
#pragma unroll
      for (int k = 0; k < UPT; ++k) {
        const int u = tid + k * 512;
        const int T = u / QPT;
        const float* zp = Zx + T * ZP + cq * 4;
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(zp);
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(zp + 64 * ZP);
        const f32x4 z2 = *reinterpret_cast<const f32x4*>(zp + 128 * ZP);
        const f32x4 z3 = *reinterpret_cast<const f32x4*>(zp + 192 * ZP);
        f32x4 ya = (z0 + z1 + z2) * sc4 + sh4;
        f32x4 yb = (z1 - z2 - z3) * sc4 + sh4;
        if (d.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ya[e] = fmaxf(ya[e], 0.f), yb[e] = fmaxf(yb[e], 0.f);
        }
        const int oy = y0 + 2 * (T >> 4), ox = x0 + 2 * (T & 15) + q;
        if (d.stat_slots) {   // BatchNorm batch statistics of the train-mode forward ride in the epilogue (no extra pass over z)
          const float ma = (interior || (ox < d.W && oy < d.H)) ? 1.f : 0.f;
          const float mb = (interior || (ox < d.W && oy + 1 < d.H)) ? 1.f : 0.f;
          st1 += ma * ya + mb * yb;
          st2 += ma * ya * ya + mb * yb * yb;
        }
        const unsigned idx = (unsigned)((oy * d.W + ox) * d.ldout + n0);
        if (interior) {
          *reinterpret_cast<f32x4*>(img_out + idx) = ya;
          *reinterpret_cast<f32x4*>(img_out + idx + sW) = yb;
        } else if (ox < d.W) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < d.N) {
              if (oy < d.H) img_out[idx + e] = ya[e];
              if (oy + 1 < d.H) img_out[idx + sW + e] = yb[e];
            }
        }
        if (d.pool) {
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(ya[e], yb[e]);
          if (q == 0) {
            pmax[k] = m;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], pmax[k][e]);
            const int py = oy >> 1, px = ox >> 1;   // floor semantics of MaxPool2d: windows fully inside the image
            if (oy + 1 < d.H && ox < d.W) {
              float* pp = pool_out + (size_t)(py * (d.W >> 1) + px) * d.ldpool + n0;
              if (fast_n) {
                *reinterpret_cast<f32x4*>(pp) = m;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (n0 + e < d.N) pp[e] = m[e];
              }
            }
          }
        }
      }
      lds_barrier();   // Zx is rewritten by the next pass / patch
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][mi][r] = 0.f;
  }
  if (d.stat_slots) {
    // fold the per-thread sums of a channel quad (512 / QPT threads each) through LDS, then one double atomic per channel
    // and sum into row (workgroup) of the accumulator table [STAT_ROWS][2 * N] that bn_finalize_slots_kernel folds
    float* red = smem;   // [512][8]
    lds_barrier();
    *reinterpret_cast<f32x4*>(red + tid * 8) = st1;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = st2;
    lds_barrier();
    if (tid < 8 * QPT) {   // thread -> (which sum, channel quad, element)
      const int which = tid / (4 * QPT), rem = tid - which * 4 * QPT, qd = rem >> 2, e = rem & 3;
      double sum = 0.0;
      for (int k = qd; k < 512; k += QPT) sum += (double)red[k * 8 + which * 4 + e];
      const int n = nblock * NC + qd * 4 + e;
      if (n < d.N) atomicAdd(d.stat_slots + (size_t)(blockIdx.x % STAT_ROWS) * 2 * d.N + which * d.N + n, sum);
    }
  }
}

// =================================================================================================================
// Component-pair work split for the three-piece operand mode (the default): wave (i, jp) owns transform row i and the
// component PAIR j in {2 jp, 2 jp + 1} of BOTH m tiles and ALL n tiles of the workgroup.
//
// In wino3x3_f32_kernel<0, 1> wave (i, g) owns all four components of n tile g, so every V value is formed -- transformed and
// split into three bf16 pieces, 8 VALU per value -- by the two waves g = 0, 1: 490 VALU per 48 MFMAs per wave and chunk,
// 14 VALU instructions issued per MFMA (rocprofv3, profiles/r01_k_pmc3_prec1.txt), and the matrix pipe 24 % busy.  Here a
// value is formed exactly once per workgroup: a wave needs only three of the four columns of its raw rows (components 0, 1
// <- columns 0..2; components 2, 3 <- columns 1..3: 6 instead of 8 LDS reads per half) and splits 32 values instead of 64
// per chunk: ~270 VALU per 48 MFMAs.  The same 128 accumulator and 48 weight-piece registers as before.
// The two component pairs of a row meet in the inverse transform: Z[i][q] = sum_j M[i][j] A[j][q] is a sum over j, so the
// jp = 0 waves store their part of it into the exchange buffer and the jp = 1 waves add theirs with ds_add_f32 one barrier
// later; the finishing pass (row part, epilogue, fused pool / statistics) is the one of wino3x3_f32_kernel.
// =================================================================================================================
// floats of one raw halo buffer of wino3x3_cp_kernel (10 x 34 pixels x 20, staged in 3 x 128 slots) = its exchange buffer
constexpr int WINO_CP_RAWF = 8192;

// DEEP (narrow layers, even chunk count): the raw halo and the weight pieces are requested TWO chunks ahead instead of one.
// A chunk of a 32-channel tile is only 24 MFMAs per wave (~0.7 us), less than an HBM round trip under load, so with one chunk
// of lead every chunk of the 512^2 layers waited for its halo (2.5-2.8 TB/s, neither pipe busy).  The memory pipe retires a
// wave's loads in order, so BOTH kinds of load need the longer lead: a weight piece issued behind a young halo load could not
// complete before it.  Costs a second set of halo (12) and weight-piece (24) registers; chunk parity is static (loop unrolled
// by two), so no register is ever moved.
// URES (DEEP with exactly two chunks, i.e. 32 input channels): the two register sets of weight pieces hold the layer's WHOLE
// transformed filter slice of this wave, so they are loaded once per workgroup and never again -- no weight-piece load sits in
// the in-order memory pipe behind the halo loads and the epilogue's stores.
#if defined(MGU_DIAG) && MGU_DIAG == 20
// diagnostic build: phase timeline of ONE workgroup of the selected layer (-DMGU_DIAG_H=.. -DMGU_DIAG_CP=.. -DMGU_DIAG_N=..), every wave's
// lane 0 stamping s_memtime at the phase boundaries of every patch: mgu_diag_ts[wave][patch][slot]; tools/diag_timeline.py reads it
#ifndef MGU_DIAG_H
#define MGU_DIAG_H 512
#endif
#ifndef MGU_DIAG_CP
#define MGU_DIAG_CP 32
#endif
#ifndef MGU_DIAG_N
#define MGU_DIAG_N 32
#endif
__device__ unsigned long long mgu_diag_ts[8][64][32];
#define DIAG_T(slot)                                                                                                       \
  do {                                                                                                                     \
    if (diag_on && (threadIdx.x & 63) == 0 && pi < 64) mgu_diag_ts[threadIdx.x >> 6][pi][slot] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define DIAG_T(slot) do {} while (0)
#endif
template <int NTB, bool STATS, bool DEEP, bool URES = false>
__global__ __launch_bounds__(512) void wino3x3_cp_kernel(const IgemmDesc d, const int tiles_x, const int tiles_y, const int total_patches,
                        const int patches_per_block, const int ngroups, const int nitems, const int per_xcd, const int flags) {
  constexpr int NWAVES = 8;
  constexpr int MT = 2;                          // both m tiles of the 8 x 32 pixel patch
  const int yfast = flags & 1;
  // flags & 2 (MGU_WINO_PRIO=1, A/B): static priority for the second-dispatched half of the workgroup (waves 4-7 = the jp = 1
  // waves, each the SIMD partner of wave - 4): the younger wave loses every VALU arbitration against its partner otherwise
  // (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
  if ((flags & 2) && (threadIdx.x >> 8)) __builtin_amdgcn_s_setprio(1);
  constexpr int NC = 32 * NTB;                   // output channels per workgroup
  constexpr int ZP = 32;                         // exchange-buffer pitch of a tile (floats): one n tile per pass, no padding
                                                 // (32 lanes write 32 consecutive floats; the finishing 16-byte reads of
                                                 // thread (tile T = t / 8, quad t % 8) fall on 64 distinct banks per lane group)
  constexpr int RH = 10, RW = 34, HPIX = RH * RW;
  constexpr int PLD = 20;
  constexpr int HSTRIDE = NWAVES * 16;
  constexpr int HR = (HPIX + HSTRIDE - 1) / HSTRIDE;
  constexpr int S1 = 17 * PLD, S2 = PLD, S3 = 17 * PLD + PLD;   // LDS offsets of tile columns 1..3 (parity planes)
  // A raw buffer also serves as the exchange buffer of the inverse transform ([4 rows i][64 tiles][32 channels] = 8192 floats):
  // at the end of a patch the buffer of the chunk just consumed is free (the other one already holds the next patch's first
  // chunk) and is the first of the four exchange regions; three more follow the raw buffers (epilogue comment).  One workgroup
  // per CU either way: the 128 / 64 accumulators plus operands of eight waves fill the register file.
  constexpr int RAWF = WINO_CP_RAWF;
  static_assert(RAWF >= HR * HSTRIDE / RW * 34 * PLD + 34 * PLD && RAWF >= 4 * 64 * ZP, "raw buffer too small");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // LDS map (five 32 KB slots): raw buffer 1 | exchange regions 1..3 | raw buffer 0.  Raw buffer b lives at smem + (b ^ 1) * BUFSTEP:
  // the exchange writes use ds_write_addtid_b32, whose M0 + offset addressing ends at 131070 bytes, so the top slot must be the
  // buffer that is never an exchange region -- buffer 0 when a patch has an even number of chunks (every layer of the U-Net: the
  // first chunk of every patch then sits in buffer 0); with an odd chunk count the consumed buffer alternates and the shares that go
  // to the top slot take ordinary ds_write_b32 stores.
  constexpr int BUFSTEP = 4 * RAWF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, jp = wave >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  const int tx = lr & 15, ty = lr >> 4;
  const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (item >= nitems) return;
  const int nblock = item / ngroups;
  const int p_begin = (item - nblock * ngroups) * patches_per_block;
  const int npatch = min(patches_per_block, total_patches - p_begin);
  if (npatch <= 0) return;
  // row i of B^T d:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float sgn = wi == 1 ? 1.f : -1.f;
  int offA[MT], offB[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int rowbase = 2 * (2 * mi + ty);
    offA[mi] = ((rowbase + ra) * 34 + tx) * PLD + lh * 8;
    offB[mi] = ((rowbase + rb) * 34 + tx) * PLD + lh * 8;
  }
  const int nC = d.Cp >> 4;                        // 16-channel raw chunks
  // three-piece layout: 16-byte lane loads, [n tile][chunk][i*4+j][piece][lane]
  // weight pieces through a buffer descriptor: wave-uniform base + SGPR offset + ONE 32-bit lane offset (lane * 16 bytes), so
  // the 12 NTB piece loads of a chunk cost no 64-bit VGPR address arithmetic and no address registers
  const size_t u_base = ((size_t)(nblock * NTB) * nC * 16 + wi * 4 + 2 * jp) * 192 * sizeof(u32x4);
  const size_t u_bytes = (size_t)NTB * nC * 16 * 192 * sizeof(u32x4);
  const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(d.wu)) + u_base, 0,
                                                         (int)u_bytes, 0x00020000);
  const unsigned nt_stride = (unsigned)nC * 16 * 192 * (unsigned)sizeof(u32x4);
  const unsigned lane16 = (unsigned)lane * 16u;
  // ---- raw halo staging (as wino3x3_f32_kernel) ----
  const int kq = tid & 3, hp0 = tid >> 2;
  int hoff[HR];
  unsigned hmask = 0u, hmask_next = 0u;
  auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.in), 0, 0x7ffffff0, 0x00020000);
  const int img_bytes = d.H * d.W * d.ldin * (int)sizeof(float);   // the launcher guarantees H*W*ldin < 2^31 elements ... and bytes fit below
  // Patch order inside an image: x fastest by default.  y fastest (MGU_WINO_YFAST=1) lets vertically adjacent patches, which
  // share two of their ten halo rows, follow each other while those rows are still in the XCD's L2 -- measured with alternating
  // runs on one box: 1 % SLOWER on the headline step (the HBM read volume is not what bounds these kernels).
  auto setup_patch = [&](int p, int& img, int& y0, int& x0) {
    const int py = yfast ? p % tiles_y : (p / tiles_x) % tiles_y;
    const int px = yfast ? (p / tiles_y) % tiles_x : p % tiles_x;
    img = p / (tiles_x * tiles_y);
    y0 = py * 8;
    x0 = px * 32;
  };
  auto setup_load = [&](int p) {
    int img, y0, x0;
    setup_patch(p, img, y0, x0);
    // wave-uniform image base in the descriptor; the lane's pixel and 16-byte piece ride in hoff (bytes)
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.in) + (size_t)img * d.H * d.W * d.ldin, 0, img_bytes, 0x00020000);
    unsigned mk = 0u;
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HSTRIDE * i;
      const int r = hp / RW, cc = hp - r * RW;
      const int y = y0 - 1 + r, x = x0 - 1 + cc;
      const bool ok = hp < HPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
      // a pixel outside the image gets an offset past the descriptor's range: the buffer load returns zeros by itself (no mask,
      // no select when the registers are written to LDS)
      // (narrow kernel only: in the wide one, at the 256-register limit, the change moved the allocation from 3 to 10 spilled registers
      // and cost 12 %)
      hoff[i] = (NTB == 1 && !ok) ? 0x7fff0000 : ((ok ? (y * d.W + x) * d.ldin : 0) + kq * 4) * (int)sizeof(float);
      mk |= ok ? (1u << i) : 0u;
    }
    hmask_next = mk;
  };
  constexpr int NSET = DEEP ? 2 : 1;   // register sets of halo / weight pieces in flight (set of chunk g = g & 1 when DEEP)
  f32x4 hreg[NSET][HR];
  unsigned hmask_set[NSET];
  (void)hmask;
  auto load_halo = [&](int c, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    hmask_set[set] = hmask_next;
#pragma unroll
    for (int i = 0; i < HR; ++i)
#if defined(MGU_DIAG) && MGU_DIAG == 3   // diagnostic build: no halo loads
      hreg[set][i] = f32x4{1.f, 1.f, 1.f, 1.f};
#else
#if defined(MGU_DIAG) && MGU_DIAG == 30   // diagnostic build (timing only): every chunk loads the channels of chunk c & 1 -- after the first
      // pair of a pixel every halo load hits in L2: what the kernel would take if no chunk waited for an HBM miss
      hreg[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, hoff[i], (c & 1) * 64, 0));
#else
      hreg[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, hoff[i], c * 64, 0));
#endif
#endif
  };
  auto store_halo = [&](float* Hs, auto set_c) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HSTRIDE * i;
      const int r = hp / RW, cc = hp - r * RW;
      *reinterpret_cast<f32x4*>(Hs + ((r * 2 + (cc & 1)) * 17 + (cc >> 1)) * PLD + kq * 4) =
          (NTB == 1 || ((hmask_set[set] >> i) & 1u)) ? hreg[set][i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int lp = 0, lc = 0;
  auto prep_next = [&]() {
    if (lc == 0 && lp < npatch) setup_load(p_begin + lp);
  };
  auto load_next = [&](auto set_c) {
    load_halo(lc, set_c);
    lc = lc + 1 == nC ? 0 : lc + 1;
    lp += lc == 0 ? 1 : 0;
  };

  const bool fast_n = (d.N % NC == 0) && (d.ldout % 4 == 0) && (d.coff % 4 == 0) && (!d.pool || d.ldpool % 4 == 0);   // uniform
  u32x4 bx[NSET][2][NTB][3];   // the three pieces of this wave's two components, every n tile, one 16-channel chunk (per set)
  auto load_bx = [&](int jj, int chunk, auto set_c) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        bx[set][jj][nt][pc] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            u_rsrc, lane16, (unsigned)(nt * nt_stride + ((unsigned)chunk * (16 * 192) + jj * 192 + pc * 64) * 16u), 0));
  };

  f32x16 acc[2][NTB][MT];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][nt][mi][r] = 0.f;

  // A chunk is four STEPS (component jj of the pair, m tile mi), 6 NTB MFMAs each.  Component jj needs TWO raw columns:
  //   V = (ra_x + sgn rb_x) + w (ra_y + sgn rb_y)
  //   jp = 0: V0 = r0 - r2 (x = col 0, y = col 2, w = -1),  V1 = r1 + r2 (x = col 1, y = col 2, w = +1)
  //   jp = 1: V2 = r2 - r1 (x = col 2, y = col 1, w = -1),  V3 = r1 - r3 (x = col 1, y = col 3, w = -1)
  u32x4 pc[2][3];   // the three bf16 pieces of a step's A operand, double buffered
  // RPF (narrow layers): the raw LDS operands of step k + 2 are requested during step k and transformed during step k + 1, so
  // no LDS round trip sits between a read and the VALU work that consumes it.  With only 6 MFMAs per step a wave has ~190 cycles
  // of MFMA time to hide behind, and the one-step pipeline (reads and their transform in the same step) left ~6 exposed
  // lgkmcnt(0) waits per step: the 32-channel layers ran the VALU a third of the time.  Costs 32 registers (a second raw set).
  constexpr bool RPF = NTB == 1;
  f32x4 raw[RPF ? 2 : 1][2][4];   // [set][hf][ax, bx, ay, by]
  auto fetch_raw = [&](const float* Hs, const int jj, const int mi, const int set) {
    const int cx = jj == 0 ? (jp ? S2 : 0) : S1;
    const int cy = jj == 0 ? (jp ? S1 : S2) : (jp ? S3 : S2);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float* pa = Hs + offA[mi] + hf * 4;
      const float* pb = Hs + offB[mi] + hf * 4;
      raw[set][hf][0] = *reinterpret_cast<const f32x4*>(pa + cx);
      raw[set][hf][1] = *reinterpret_cast<const f32x4*>(pb + cx);
      raw[set][hf][2] = *reinterpret_cast<const f32x4*>(pa + cy);
      raw[set][hf][3] = *reinterpret_cast<const f32x4*>(pb + cy);
    }
  };
  auto form_from = [&](const int jj, const int set, const int slot) {
    const float w = (jj == 1 && jp == 0) ? 1.f : -1.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      // element by element with scalar FMAs: on <4 x float> values the backend selects v_pk_fma_f32, and a packed fp32
      // instruction beside the MFMAs costs ~22 cycles more than the two v_fma_f32 it replaces (MI355X_MICROARCH.md)
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float qx = scalar_fma(sgn, raw[set][hf][1][e], raw[set][hf][0][e]);
        const float qy = scalar_fma(sgn, raw[set][hf][3][e], raw[set][hf][2][e]);
        v[e] = scalar_fma(w, qy, qx);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        unsigned p0, p1, p2;
        split3_pack(v[2 * e], v[2 * e + 1], p0, p1, p2);
        pc[slot][0][hf * 2 + e] = p0, pc[slot][1][hf * 2 + e] = p1, pc[slot][2][hf * 2 + e] = p2;
      }
    }
  };
  auto form = [&](const float* Hs, const int jj, const int mi, const int slot) {
    fetch_raw(Hs, jj, mi, 0);
    form_from(jj, 0, slot);
  };

  // Issue order = the steady state's (vmcnt retires in order and hipcc merges the pending-load state of the loop's two
  // predecessors): the oldest pending loads at the top of a chunk are the halo of the next chunk, then the weight pieces.
  using S0 = std::integral_constant<int, 0>;
  using S1c = std::integral_constant<int, DEEP ? 1 : 0>;
  prep_next();
  load_next(S0{});
  store_halo(smem + BUFSTEP, S0{});
  prep_next();
  load_next(S1c{});
  load_bx(0, 0, S0{});
  load_bx(1, 0, S0{});
  if constexpr (DEEP) {   // chunk 2 -> halo set 0, chunk 1 -> weight set 1 (nC is even, >= 2)
    prep_next();
    load_next(S0{});
    load_bx(0, 1, S1c{});
    load_bx(1, 1, S1c{});
  }
  lds_barrier();
  form(smem + BUFSTEP, 0, 0, 0);      // step 0 of the first chunk
  if constexpr (RPF) fetch_raw(smem + BUFSTEP, 0, 1, 1);   // raw operands of step 1 (set = step & 1)
  f32x4 st1[NTB], st2[NTB];   // (STATS) per-thread sums of its channel quad of every n tile
#pragma unroll
  for (int nt = 0; nt < NTB; ++nt) st1[nt] = f32x4{0.f, 0.f, 0.f, 0.f}, st2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int buf = 0;
#if defined(MGU_DIAG) && MGU_DIAG == 20
  const bool diag_on = blockIdx.x == 100 && blockIdx.y == 0 && d.H == MGU_DIAG_H && d.Cp == MGU_DIAG_CP && d.N == MGU_DIAG_N;
#endif
  for (int pi = 0; pi < npatch; ++pi) {
    DIAG_T(0);
    auto chunk_body = [&](const int c, auto par_c) {
      constexpr int P = decltype(par_c)::value;                  // DEEP: parity of the chunk = its register set
      using SetNext = std::integral_constant<int, DEEP ? (P ^ 1) : 0>;   // set holding chunk c + 1 (stored now, then refilled)
      using SetCur = std::integral_constant<int, DEEP ? P : 0>;          // weight pieces of this chunk
      // Software pipeline WITHOUT a lead-in or a tail: the MFMAs of every step run beside the LDS reads, transform and
      // three-way split of the NEXT step -- and the last step's partner is step 0 of the next chunk, whose raw data (parked
      // in the other buffer at the top of this chunk) become visible at the mid-chunk barrier B1.  So a wave's VALU work
      // always has MFMAs of its own to hide behind (the bf16 MFMA lets the VALU issue beside it), and the two waves of a SIMD
      // (the two component pairs of a row) need not alternate phases.  Two barriers per chunk:
      //   B0 (top): every wave has finished reading the buffer that now receives chunk c + 1;
      //   B1 (after step 1): chunk c + 1 is complete in LDS.
      prep_next();
      lds_barrier();                                     // B0
      if (c < 4) DIAG_T(1 + 3 * c);
      const float* Hs = smem + (buf ^ 1) * BUFSTEP;      // chunk c
      const float* Hn = smem + buf * BUFSTEP;            // chunk c + 1 (or the next patch's first)
      store_halo(smem + buf * BUFSTEP, SetNext{});
      load_next(SetNext{});                              // DEEP: chunk c + 3 (else c + 2)
      const int cn = DEEP ? (c + 2 >= nC ? c + 2 - nC : c + 2) : (c + 1 == nC ? 0 : c + 1);   // chunk whose weight pieces are requested
      static_for<0, 4>([&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
        constexpr int jj = st >> 1, mi = st & 1, slot = st & 1;
        if constexpr (st == 2) {
          if (c < 4) DIAG_T(2 + 3 * c);                  // arrival at B1
          lds_barrier();                                 // B1
          if (c < 4) DIAG_T(3 + 3 * c);
        }
#if defined(MGU_DIAG) && MGU_DIAG == 11   // diagnostic build: no MFMAs in the chunk loop (operands folded into one register each)
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
          for (int q3 = 0; q3 < 3; ++q3)
            acc[jj][nt][mi][q3] += __uint_as_float(pc[slot][q3][0] ^ pc[slot][q3][1] ^ pc[slot][q3][2] ^ pc[slot][q3][3] ^ bx[SetCur::value][jj][nt][q3][0]);
#else
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
          f32x16 t = acc[jj][nt][mi];
          t = mfma_bf16(pc[slot][2], bx[SetCur::value][jj][nt][0], t);
          t = mfma_bf16(pc[slot][0], bx[SetCur::value][jj][nt][2], t);
          t = mfma_bf16(pc[slot][1], bx[SetCur::value][jj][nt][1], t);
          t = mfma_bf16(pc[slot][1], bx[SetCur::value][jj][nt][0], t);
          t = mfma_bf16(pc[slot][0], bx[SetCur::value][jj][nt][1], t);
          t = mfma_bf16(pc[slot][0], bx[SetCur::value][jj][nt][0], t);
          acc[jj][nt][mi] = t;
        }
#endif
#if defined(MGU_DIAG) && MGU_DIAG == 10   // diagnostic build: no input transform / split in the chunk loop (pieces stay constant)
        if constexpr (false) {
#else
        if constexpr (RPF) {
#endif
          // transform the raw operands of step st + 1 (read one step ago), request those of step st + 2 (the next chunk's after
          // B1: steps 2 and 3 read chunk c + 1)
          form_from(((st + 1) >> 1) & 1, (st + 1) & 1, slot ^ 1);
          if constexpr (st < 2) fetch_raw(Hs, (st + 2) >> 1, (st + 2) & 1, st & 1);
          else fetch_raw(Hn, (st - 2) >> 1, (st - 2) & 1, st & 1);
        } else {
#if !(defined(MGU_DIAG) && MGU_DIAG == 10)
          if constexpr (st < 3) form(Hs, (st + 1) >> 1, (st + 1) & 1, slot ^ 1);
          else form(Hn, 0, 0, slot ^ 1);
#endif
        }
        if constexpr (mi == 1 && !URES) load_bx(jj, cn, SetCur{});   // this component's pieces of the next chunk using this set
        constexpr int NM = 6 * NTB;                               // MFMAs of the step
        constexpr int per = (72 + NM - 1) / NM;                   // transform (24) + split (44) + addresses of the next step
#pragma unroll
        for (int k = 0; k < NM; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA
          if (NM >= 8 ? (k < 8) : true) __builtin_amdgcn_sched_group_barrier(0x100, NM >= 8 ? 1 : 2, 0);   // LDS reads (8 per step)
          __builtin_amdgcn_sched_group_barrier(0x002, per, 0);                                 // VALU
          if constexpr (mi == 1 && !URES) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a weight-piece load
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      buf ^= 1;
    };
    if constexpr (DEEP) {
      for (int c = 0; c < nC; c += 2) {
        chunk_body(c, std::integral_constant<int, 0>{});
        chunk_body(c + 1, std::integral_constant<int, 1>{});
      }
    } else {
      for (int c = 0; c < nC; ++c) chunk_body(c, std::integral_constant<int, 0>{});
    }
    // ---- inverse transform + epilogue of patch pi ----
    int img, y0, x0;
    setup_patch(p_begin + pi, img, y0, x0);
    float* const img_out = d.out + (size_t)img * d.H * d.W * d.ldout + d.coff;
    const unsigned sW = (unsigned)(d.W * d.ldout);
    const bool interior = (y0 + 8 <= d.H) && (x0 + 32 <= d.W) && fast_n;
    float* pool_out = nullptr;
    if (d.pool) pool_out = d.pool + (size_t)img * (d.H >> 1) * (d.W >> 1) * d.ldpool;
    // Everything per-lane the epilogue needs is re-derived here from an OPAQUE copy of the thread id, so that none of it is
    // live across the main loop (hipcc otherwise hoists the loop-invariant exchange-buffer addresses out of the patch loop,
    // keeps them live through the main loop at the register limit and spills them: each reload then waited vmcnt(0) in the
    // middle of the epilogue, 2.2x on the whole kernel).
    int et = threadIdx.x;
    asm volatile("" : "+v"(et));
    const int cq = et & 7, T = et >> 3;              // finishing unit of this thread: (tile, channel quad of the pass's n tile)
    // Slot of tile T inside a region row: the writers store register r of m tile mi at slot 32 mi + 2 r + h (h = lane >> 5: the two
    // half waves of a store are adjacent 128-byte rows, the add-TID form), and register r of half h is tile (r & 3) + 8 (r >> 2) + 4 h.
    const int Tslot = (T & 32) + 2 * ((T & 3) + 4 * ((T & 31) >> 3)) + ((T >> 2) & 1);
    // Per-channel scale / shift of the finishing unit's channel quad (y = scale * (Z0 + Z1 + Z2) + shift: 16 FMAs per unit; on the
    // writer side -- every share scaled by its lane's channel, as rounds 2-3 had it so that the finishing pass needed no per-channel
    // data -- it was 64 per thread, and with the add-TID stores the share arithmetic is what the write phase takes).  Loaded per patch
    // (not held through the main loop), in front of the epilogue's first barrier (the barrier wait covers the L2 round trip),
    // unconditionally (clamped index) and BEFORE the first output store of the patch: the wait for a load is a wait for every
    // older memory operation of the wave (vmcnt retires in order), so a load -- or a scratch reload of one -- behind the
    // previous pass's stores waits for their HBM round trip (measured: the whole gain of the barrier-light exchange).
    // Inference variants: the train-mode (STATS) variants keep the writer-side form -- they sit at 256 registers with spills, and the
    // eight extra live registers per n tile add to those (STATS: 72 -> 136 bytes of scratch per lane).  The wide inference kernel
    // pays 12 -> 28 bytes of scratch and +0.6 % per launch for it in THIS C++ form, but it is the fallback now: the assembly kernel
    // (csrc/asm/gen_wino_cp.py) that runs these layers has the registers (the dead accumulators) and saves 96 VALU per patch and wave
    // with the reader-side form, and the two must stay bit-for-bit equal.
    constexpr bool RSC = !STATS;
    f32x4 sc4[NTB], sh4[NTB];
    float scw[NTB], shw[NTB];
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
      const int n0 = nblock * NC + nt * 32 + cq * 4;
      sc4[nt] = f32x4{1.f, 1.f, 1.f, 1.f}, sh4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      scw[nt] = 1.f, shw[nt] = 0.f;
      if constexpr (!RSC) {
        const int n = min(nblock * NC + nt * 32 + (et & 31), d.N - 1);
        scw[nt] = d.scale ? d.scale[n] : 1.f;
        shw[nt] = (d.shift && wi == 1 && jp == 0) ? d.shift[n] : 0.f;   // Z1 enters both output rows with +: the shift is added once
      } else if (fast_n) {
        if (d.scale) sc4[nt] = *reinterpret_cast<const f32x4*>(d.scale + n0);
        if (d.shift) sh4[nt] = *reinterpret_cast<const f32x4*>(d.shift + n0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = min(n0 + e, d.N - 1);
          if (d.scale) sc4[nt][e] = d.scale[n];
          if (d.shift) sh4[nt][e] = d.shift[n];
        }
      }
    }
    DIAG_T(13);      // main loop done (arrival at the epilogue's first barrier)
    lds_barrier();   // every wave has finished reading the consumed raw buffer, which is exchange region 0 from here on
    DIAG_T(14);
    // Exchange regions Z[q][jp] ([4 rows i][64 tiles][32 channels] each): the column part of Z[i][q] = sum_j M[i][j] A[j][q]
    // (A^T = [1 1 1 0; 0 1 -1 -1]) is split over the two component-pair waves of a row,
    //   jp = 0 (M0, M1): q = 0: M0 + M1, q = 1: M1;      jp = 1 (M2, M3): q = 0: M2, q = 1: -M2 - M3,
    // and every wave writes its two shares to regions of its OWN; the finishing pass adds the two shares while it reads.  (An
    // earlier version had the jp = 1 waves read-add-write the jp = 0 waves' region: a serialised LDS round trip per group of
    // four slots with half the waves idle, and three barriers per (q, n tile) pass -- a quarter of the whole kernel's time.)
    // Region (0, 0) is the raw buffer just consumed (buffer buf holds the next patch's first chunk), regions (0, 1), (1, 0), (1, 1) are
    // slots 1..3 of the LDS map: 64 KB + 96 KB = the CU's 160 KB.
    float* const zreg0 = smem + buf * BUFSTEP;
    auto zregion = [&](const int q, const int j) { return (q | j) == 0 ? zreg0 : smem + (2 * q + j) * RAWF; };
    // add-TID bases of this wave's two shares (row wi of the region): the q = 0 regions lie below 64 KB (offset bias 0), the q = 1
    // regions in [64 KB, 128 KB) (offset bias ZBIAS, so that M0 = address - ZBIAS stays within 16 bits)
    constexpr int ZBIAS = 57472;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const bool top0 = jp == 0 && buf != 0;   // this wave's q = 0 share goes to the top slot (odd chunk counts only): ordinary stores
    const unsigned mz0 = lds0 + (unsigned)((jp == 0 ? buf * BUFSTEP : RAWF) + wi * 64 * ZP) * 4u;
    const unsigned mz1 = lds0 + (unsigned)((2 + jp) * RAWF + wi * 64 * ZP) * 4u - (unsigned)ZBIAS;
#if defined(MGU_DIAG) && MGU_DIAG == 2   // diagnostic build: no inverse transform / epilogue at all
    {
      float sacc = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[jj][nt][mi][r];
      if (sacc == 123.456f) img_out[et] = sacc;
    }
#else
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
      const int n0 = nblock * NC + nt * 32 + cq * 4;
      {
        // the wave's role jp is uniform: one branch around two specialised copies instead of a select per value (1 instead of 4
        // VALU per register pair; with the add-TID stores the share arithmetic is what this phase takes)
        auto write_shares = [&](auto JP0) {
          constexpr bool jp0 = decltype(JP0)::value;
          auto share = [&](const int q, const int mi, const int r) {
            const float m0 = acc[0][nt][mi][r], m1 = acc[1][nt][mi][r];
            if constexpr (RSC) return q == 0 ? (jp0 ? m0 + m1 : m0) : (jp0 ? m1 : -m0 - m1);   // two of the four shares are plain copies
            else return q == 0 ? (jp0 ? m0 + m1 : m0) * scw[nt] + shw[nt] : (jp0 ? m1 : -m0 - m1) * scw[nt] + shw[nt];
          };
          if (!(jp0 && top0)) {
            static_for<0, MT * 4>([&](auto G) {
              constexpr int mi = decltype(G)::value >> 2, r = 4 * (decltype(G)::value & 3);
              lds_store4_addtid<(32 * mi + 2 * r) * ZP * 4>(mz0, share(0, mi, r), share(0, mi, r + 1), share(0, mi, r + 2), share(0, mi, r + 3));
            });
          } else {
            float* const z0p = zreg0 + (wi * 64 + ((et >> 5) & 1)) * ZP + (et & 31);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) z0p[(32 * mi + 2 * r) * ZP] = share(0, mi, r);
          }
          static_for<0, MT * 4>([&](auto G) {
            constexpr int mi = decltype(G)::value >> 2, r = 4 * (decltype(G)::value & 3);
            lds_store4_addtid<ZBIAS + (32 * mi + 2 * r) * ZP * 4>(mz1, share(1, mi, r), share(1, mi, r + 1), share(1, mi, r + 2), share(1, mi, r + 3));
          });
        };
        if (jp == 0) write_shares(std::true_type{});
        else write_shares(std::false_type{});
      }
      if (nt == 0) DIAG_T(15);   // shares written (arrival)
      lds_barrier();
      if (nt == 0) DIAG_T(16);
#if defined(MGU_DIAG) && MGU_DIAG == 5   // diagnostic build: exchange writes and barriers only
      lds_barrier();
      continue;
#endif
      // row part + epilogue of unit (T, cq): y(2tr, .) = Z0 + Z1 + Z2, y(2tr+1, .) = Z1 - Z2 - Z3, both output columns q
      f32x4 ya[2], yb[2];
      // all sixteen share reads of the unit in flight before the first add (the epilogue has the registers; two waves per SIMD do not
      // cover an LDS round trip per output column)
      f32x4 zs[2][2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) zs[q][j][i] = *reinterpret_cast<const f32x4*>(zregion(q, j) + (i * 64 + Tslot) * ZP + cq * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 z0 = zs[q][0][0] + zs[q][1][0];
        const f32x4 z1 = zs[q][0][1] + zs[q][1][1];
        const f32x4 z2 = zs[q][0][2] + zs[q][1][2];
        const f32x4 z3 = zs[q][0][3] + zs[q][1][3];
        ya[q] = z0 + z1 + z2;
        yb[q] = z1 - z2 - z3;
        if constexpr (RSC) ya[q] = ya[q] * sc4[nt] + sh4[nt], yb[q] = yb[q] * sc4[nt] + sh4[nt];
        if (d.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ya[q][e] = relu_1op(ya[q][e]), yb[q][e] = relu_1op(yb[q][e]);
        }
      }
      const int oy = y0 + 2 * (T >> 4), ox = x0 + 2 * (T & 15);
      if (STATS) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float ma = (interior || (ox + q < d.W && oy < d.H)) ? 1.f : 0.f;
          const float mb = (interior || (ox + q < d.W && oy + 1 < d.H)) ? 1.f : 0.f;
          st1[nt] += ma * ya[q] + mb * yb[q];
          st2[nt] += ma * ya[q] * ya[q] + mb * yb[q] * yb[q];
        }
      }
      const unsigned idx = (unsigned)((oy * d.W + ox) * d.ldout + n0);
#if defined(MGU_DIAG) && MGU_DIAG == 1   // diagnostic build: the output stores only if a value is a magic number
      if (ya[0][0] == 123.456f) {
#else
      if (interior) {
#endif
        // write-once data that the next layer reads after this kernel has finished: non-temporal (streaming) stores, measured
        // 1.2 % of the headline step against ordinary stores
        __builtin_nontemporal_store(ya[0], reinterpret_cast<f32x4*>(img_out + idx));
        __builtin_nontemporal_store(ya[1], reinterpret_cast<f32x4*>(img_out + idx + d.ldout));
        __builtin_nontemporal_store(yb[0], reinterpret_cast<f32x4*>(img_out + idx + sW));
        __builtin_nontemporal_store(yb[1], reinterpret_cast<f32x4*>(img_out + idx + sW + d.ldout));
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (ox + q < d.W) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + e < d.N) {
                if (oy < d.H) img_out[idx + q * d.ldout + e] = ya[q][e];
                if (oy + 1 < d.H) img_out[idx + q * d.ldout + sW + e] = yb[q][e];
              }
          }
      }
      if (d.pool) {
        // the 2x2 output tile IS a pooling window (floor semantics: only complete windows)
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(ya[0][e], yb[0][e]), fmaxf(ya[1][e], yb[1][e]));
        const int py = oy >> 1, px = ox >> 1;
        if (oy + 1 < d.H && ox + 1 < d.W) {
          float* pp = pool_out + (size_t)(py * (d.W >> 1) + px) * d.ldpool + n0;
          if (fast_n) {
            *reinterpret_cast<f32x4*>(pp) = m;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + e < d.N) pp[e] = m[e];
          }
        }
      }
      if (nt == 0) DIAG_T(17);   // finishing pass issued (arrival)
      lds_barrier();   // the regions are rewritten by the next pass / region 0 receives the next raw chunk
      if (nt == NTB - 1) DIAG_T(18);
    }
#endif
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[jj][nt][mi][r] = 0.f;
  }
  if (STATS) {
    // fold the per-thread sums of a channel quad (64 threads each) through LDS, then one double atomic per channel and sum into
    // row (workgroup) of the accumulator table [STAT_ROWS][2 * N] that bn_finalize_slots_kernel folds (one adder per element)
    float* red = smem;   // [NTB][512][8]
    lds_barrier();
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
      *reinterpret_cast<f32x4*>(red + (nt * 512 + tid) * 8) = st1[nt];
      *reinterpret_cast<f32x4*>(red + (nt * 512 + tid) * 8 + 4) = st2[nt];
    }
    lds_barrier();
    if (tid < 64 * NTB) {   // thread -> (n tile, which sum, channel quad, element)
      const int nt = tid >> 6, rem = tid & 63, which = rem >> 5, qd = (rem >> 2) & 7, e = rem & 3;
      double sum = 0.0;
      for (int k = qd; k < 512; k += 8) sum += (double)red[(nt * 512 + k) * 8 + which * 4 + e];
      const int n = nblock * NC + nt * 32 + qd * 4 + e;
      if (n < d.N) atomicAdd(d.stat_slots + (size_t)(blockIdx.x % STAT_ROWS) * 2 * d.N + which * d.N + n, sum);
    }
  }
}

template <int NTB, bool STATS, bool DEEP = false, bool URES = false>
static hipError_t launch_wino_cp(const IgemmDesc& d, hipStream_t s) {
  constexpr int NWAVES = 8;
  const int tiles_x = (d.W + 31) / 32, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B, nblk = (d.N + 32 * NTB - 1) / (32 * NTB);
  // one workgroup per CU is resident (registers: 8 waves x ~200-256): ONE round of 256, each walking its share of the patches
  const int rounds = std::max(1, tun(d).wino_rounds), cap = std::max(1, tun(d).wino_ppb_cap);
  int ppb = (int)(((long)total * nblk) / (256 * rounds));
  if (ppb < 1) ppb = 1;
  if (ppb > cap) ppb = cap;
  const int ngroups = (total + ppb - 1) / ppb;
  const int per_xcd = (ngroups * nblk + 7) / 8;
  dim3 grid(8 * per_xcd, 1);
  if (d.stat_slots && grid.x > (unsigned)STAT_ROWS) return hipErrorInvalidValue;   // one accumulator row per workgroup (common.h)
  const size_t lds = (size_t)(5 * WINO_CP_RAWF) * sizeof(float);   // two raw buffers + three exchange regions = the CU's 160 KB
  static bool attr_done[64] = {};
  hipError_t ae = ensure_dyn_lds(reinterpret_cast<const void*>(&wino3x3_cp_kernel<NTB, STATS, DEEP, URES>), lds, attr_done);
  if (ae != hipSuccess) return ae;
  hipLaunchKernelGGL((wino3x3_cp_kernel<NTB, STATS, DEEP, URES>), grid, dim3(64 * NWAVES), lds, s, d, tiles_x, tiles_y, total, ppb, ngroups, ngroups * nblk,
                     per_xcd, (tun(d).wino_yfast ? 1 : 0) | (tun(d).wino_prio ? 2 : 0));
  return hipGetLastError();
}

template <int MODE, int PREC>
static hipError_t launch_wino_mode(const IgemmDesc& d, hipStream_t s) {
  constexpr int NTB = MODE == 0 ? 2 : 1, NWAVES = 8;
  const int tiles_x = (d.W + 31) / 32, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B, nblk = (d.N + 32 * NTB - 1) / (32 * NTB);
  // one workgroup per CU is resident: ONE round of them, each walking up to 32 patches (measured against 2-4 rounds of
  // shorter walks: fewer pipeline prologues and no second-round tail, 0.7 % of the headline step)
  const int rounds = std::max(1, tun(d).wino_rounds), cap = std::max(1, tun(d).wino_ppb_cap);
  int ppb = (int)(((long)total * nblk) / (256 * rounds));
  if (ppb < 1) ppb = 1;
  if (ppb > cap) ppb = cap;
  const int ngroups = (total + ppb - 1) / ppb;
  const int per_xcd = (ngroups * nblk + 7) / 8;
  dim3 grid(8 * per_xcd, 1);
  if (d.stat_slots && grid.x > (unsigned)STAT_ROWS) return hipErrorInvalidValue;   // one accumulator row per workgroup (common.h)
  constexpr int HSTRIDE = NWAVES * 16, HR = (340 + HSTRIDE - 1) / HSTRIDE;
  constexpr int RAWF = HR * HSTRIDE / 34 * 34 * 20 + 34 * 20;   // must match the kernel
  const size_t lds = (size_t)(2 * RAWF + 4 * 64 * (32 * NTB + 8)) * sizeof(float);
  static bool attr_done[64] = {};
  hipError_t ae = ensure_dyn_lds(reinterpret_cast<const void*>(&wino3x3_f32_kernel<MODE, PREC>), lds, attr_done);
  if (ae != hipSuccess) return ae;
  hipLaunchKernelGGL((wino3x3_f32_kernel<MODE, PREC>), grid, dim3(64 * NWAVES), lds, s, d, tiles_x, tiles_y, total, ppb, ngroups,
                     ngroups * nblk, per_xcd);
  return hipGetLastError();
}

// workgroups launch_wino_f32 starts for d (every work split uses the same formula): the rows its statistics epilogue adds into
int wino_grid_blocks(const IgemmDesc& d) {
  const bool wide = d.N > 32 && tun(d).wino_mode != 1;
  const int ntb = wide ? 2 : 1;
  const int tiles_x = (d.W + 31) / 32, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B, nblk = (d.N + 32 * ntb - 1) / (32 * ntb);
  const int rounds = std::max(1, tun(d).wino_rounds), cap = std::max(1, tun(d).wino_ppb_cap);
  int ppb = (int)(((long)total * nblk) / (256 * rounds));
  if (ppb < 1) ppb = 1;
  if (ppb > cap) ppb = cap;
  const int ngroups = (total + ppb - 1) / ppb;
  return 8 * ((ngroups * nblk + 7) / 8);
}

bool wino_applicable(const IgemmDesc& d) {
  return tun(d).use_wino && d.wu && d.KS == 3 && d.out_mode == 0 && d.split_n == 0 && (d.Cp % 16) == 0 && d.K == 9 * d.Cp &&
         (d.ldin & 3) == 0 && (long)d.H * d.W * d.ldin < (1l << 31) && (long)d.H * d.W * d.ldout < (1l << 31);
}

hipError_t launch_wino_f32(const IgemmDesc& d, hipStream_t s) {
  const bool wide = d.N > 32 && tun(d).wino_mode != 1;
  if (tun(d).wino_prec && tun(d).wino_cp && (wide || tun(d).wino_cp_narrow) &&
      (long)d.H * d.W * d.ldin * 4 < 0x7fff0000l) {   // image bytes below the out-of-image marker offset of the buffer descriptor
    // narrow layers: raw-operand prefetch always (RPF in the kernel) + the two-chunk load lead (DEEP) for an even chunk count; the
    // training forward (fused statistics) keeps the one-chunk lead (DEEP + RPF + statistics spills 4 registers)
    const bool deep = !wide && ((d.Cp >> 4) & 1) == 0 && tun(d).wino_deep && !d.stat_slots;
    const bool ures = deep && (d.Cp >> 4) == 2 && tun(d).wino_ures;
    if (d.stat_slots) return wide ? launch_wino_cp<2, true>(d, s) : deep ? launch_wino_cp<1, true, true>(d, s) : launch_wino_cp<1, true>(d, s);
    if (wino_asm_applicable(d)) return launch_wino_cp_asm(d, s);   // wide layers and the two- / four-chunk narrow layers (wino_asm.hip)
    if (deep && ures) return launch_wino_cp<1, false, true, true>(d, s);
    return wide ? launch_wino_cp<2, false>(d, s) : deep ? launch_wino_cp<1, false, true>(d, s) : launch_wino_cp<1, false>(d, s);
  }
  if (tun(d).wino_prec) return wide ? launch_wino_mode<0, 1>(d, s) : launch_wino_mode<1, 1>(d, s);
  return wide ? launch_wino_mode<0, 0>(d, s) : launch_wino_mode<1, 0>(d, s);
}

}  // namespace mgu

#if defined(MGU_DIAG) && MGU_DIAG == 20
extern "C" int mgu_diag_read(unsigned long long* out, int n) {
  if (n > 8 * 64 * 32) n = 8 * 64 * 32;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgu::mgu_diag_ts), (size_t)n * sizeof(unsigned long long));
}
#endif
