// ConvTranspose2d(kernel 2, stride 2) of the fp32 configuration (model/unet/unet_decoder.py:25,36) on the bf16 matrix cores
// with EXACT three-way operand splits -- the same arithmetic as the Winograd kernels' PREC = 1 mode (wino_f32.hip):
//   a = a0 + a1 + a2 (8 + 8 + 8 mantissa bits by truncation),  a b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a2 b0 + a1 b1),
// six v_mfma_f32_32x32x16_bf16 per 16 input channels with fp32 accumulation; the dropped terms are below one fp32 rounding.
//
// Why: the four ConvTranspose layers are one GEMM each, out[(2y+dy, 2x+dx)][co] = sum_ci in[(y,x)][ci] * w[ci][co][dy][dx]
// (M = B H W pixels, N = 4 Cout, K = Cin), 34 GFLOP per headline step.  On v_mfma_f32_32x32x2_f32 that is >= 219 us at the
// fp32 matrix peak (measured ~400 us on the generic tile kernel) against ~150 us of HBM time for the 0.75 GB they move; the
// six bf16 passes cost 6/16 of the fp32 MFMA time and leave the VALU free for the split (the fp32 MFMA blocks it).
//
// Structure: no LDS tiles and no barrier in the main loop.  Both operands reach the registers already in MFMA fragment order:
//   A: lane (r = lane & 31, h = lane >> 5) of an m tile needs in[row r][k = 16 s + 8 h .. + 7] = 32 contiguous bytes of an NHWC
//      pixel: two 16-byte global loads, split into three bf16x8 pieces in registers (5.5 VALU per value);
//   B: the weights are split and laid out per lane at load time (pack_convt_x3), so a wave's six fragments of a K = 16 step are
//      six 16-byte loads from a contiguous 6 KB block that every workgroup of the same n tile reads (L2 resident).
// Workgroup = 4 waves on a 128 pixel x 128 column tile, wave tile 64 x 64 (2 x 2 MFMA tiles, 24 MFMAs per K = 16 step),
// register double buffering one step ahead.  Workgroup ids are remapped so that the n tiles of one pixel tile run on the same
// XCD back to back: the pixel rows are fetched from HBM once and re-read from that XCD's L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "pack_small.h"

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// exact three-way split of two fp32 values into packed bf16 pieces (low half: a, high half: b)
__device__ __forceinline__ void split3_pack(const float a, const float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// weights in fragment order: pack_convt_x3_body (pack_small.h)
__global__ void pack_convt_x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ Wx, int Cin, int Cout) {
  pack_convt_x3_body(w, Wx, Cin, Cout, 0, blockIdx.x, gridDim.x);
}

// LDS hand-off barrier without the workgroup fence of __syncthreads(), which makes hipcc wait vmcnt(0): the prefetched loads of the next
// steps would be drained at every step (wino_f32.hip: lds_barrier)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 0: the forward layer.  Rows = input pixels (contiguous NHWC rows, K = Cin), columns n = (dy*2+dx)*Cout + co, pixel-shuffle store.
// MODE 1: its data gradient (loss.backward() through unet_decoder.py:36): din[(y,x)][ci] = sum_{q,co} dout[(2y+qy, 2x+qx)][co] w[ci][co][q].
//         Rows = pixels of the H x W input grid, K = 4 Cout with k = q*Cout + co: the K = 32 step s reads 32 channels of ONE of the four
//         output pixels of its row (a different base pixel per tap, same whole-row staging), columns n = ci, plain row store.
//         `Cin` is then the number of channels per tap (the layer's Cout), `Cout` the number of columns (the layer's Cin), `in` = dout.
// NI = 32-column MFMA tiles per wave: 2 (128-column workgroup tile) or 1 (64 columns: the shallowest layer's gradient has only 64).
template <int MODE, int NI>
__global__ __launch_bounds__(256, 2) void convt2x2_x3_kernel(const float* __restrict__ in, const int ldin, const uint16_t* __restrict__ Wx,
                                                             const float* __restrict__ shift, float* __restrict__ out, const int M,
                                                             const int H, const int W, const int Cin, const int Cout, const int ldout,
                                                             const int coff, const int Hout, const int Wout, const int ntn,
                                                             const int nblocks) {
  // A pieces of one K = 32 step: [2 buffers][3 pieces][128 rows][32 k + 8 pad] bf16 (80-byte rows: conflict-free ds_read_b128)
  constexpr int AP = 40, ABUF = 3 * 128 * AP;
  __shared__ __attribute__((aligned(16))) uint16_t As[2 * ABUF];
  __shared__ int rowoff[128];
  // XCD-aware order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs; logical block lb runs on XCD
  // blockIdx % 8 and a contiguous range of logical blocks (all n tiles of a pixel tile, neighbouring pixel tiles) shares an L2
  const int chunk = gridDim.x >> 3;
  const int lb = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (lb >= nblocks) return;   // block-uniform, before any barrier
  const int nt = lb % ntn, mt = lb / ntn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int bm0 = mt * 128;
  const int HW = H * W;
  // element offset of every row's output pixel (2y, 2x) relative to the tile's first one: decoded once per workgroup (two
  // integer divisions per row), read back per accumulator register
  long long pix0;
  {
    const int img = bm0 / HW, rem = bm0 - img * HW, y = rem / W;
    pix0 = ((long long)img * Hout + 2 * y) * Wout + 2 * (rem - y * W);
  }
  if (tid < 128) {
    const int m = bm0 + tid;
    int off = 0;
    if (m < M) {
      const int img = m / HW, rem = m - img * HW, y = rem / W, x = rem - y * W;
      off = (int)((((long long)img * Hout + 2 * y) * Wout + 2 * x - pix0) * ldout);   // a tile spans < 2^31 elements (host check)
    }
    rowoff[tid] = off;
  }

  // ---- A staging: the 128 x 32 fp32 tile of a step is read ONCE per workgroup in whole 128-byte rows (8 lanes x 16 B per row, 32
  // rows per instruction), split into its three bf16 pieces by the thread that loaded it, and the pieces go to LDS in the k order
  // the MFMA fragments want.  (Round 2 loaded fragment-shaped pieces straight into registers: 16-byte pieces of 32 different lines
  // per instruction, every row fetched and split by both n-halves of the workgroup, a one-step lead that did not cover an L2 round
  // trip -- the four layers ran at 20-35 % of the matrix pipe with nothing else saturated.)
  const int arow = tid >> 3, ach = (tid & 7) * 4;
  const float* aptr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = min(bm0 + j * 32 + arow, M - 1);   // rows past the end: re-read, never stored
    if (MODE == 0) {
      aptr[j] = in + (size_t)m * ldin + ach;
    } else {
      const int img = m / HW, rem = m - img * HW, y = rem / W, x = rem - y * W;
      aptr[j] = in + (((size_t)img * Hout + 2 * y) * Wout + 2 * x) * ldin + ach;   // output pixel (2y, 2x); the tap adds (qy Wout + qx) ldin
    }
  }
  const int nk = (MODE == 0 ? Cin : 4 * Cin) >> 5;   // K = 32 steps
  const int spq = Cin >> 5;                            // MODE 1: steps per tap
  auto koff = [&](int s) -> int {                      // element offset of step s's 32 channels relative to aptr (wave-uniform)
    if (MODE == 0) return s * 32;
    const int q = s / spq;
    return ((q >> 1) * Wout + (q & 1)) * ldin + (s - q * spq) * 32;
  };
  f32x4 areg[2][4];          // two steps in flight: an L2 / HBM round trip is longer than one step's 48 MFMAs
  auto load_a = [&](int s, auto set_t) {
    constexpr int set = decltype(set_t)::value;
    const int ko = koff(s);
#pragma unroll
    for (int j = 0; j < 4; ++j) areg[set][j] = *reinterpret_cast<const f32x4*>(aptr[j] + ko);
  };
  auto store_a = [&](int buf, auto set_t) {
    constexpr int set = decltype(set_t)::value;
    uint16_t* const base = As + buf * ABUF + arow * AP + ach;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned p[3][2];
      split3_pack(areg[set][j][0], areg[set][j][1], p[0][0], p[1][0], p[2][0]);
      split3_pack(areg[set][j][2], areg[set][j][3], p[0][1], p[1][1], p[2][1]);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(base + (q * 128 + j * 32) * AP) = uint2{p[q][0], p[q][1]};
    }
  };

  // a 128-column block of the packed weights holds 4 column tiles x 3 pieces per half step; a 64-column workgroup tile is half a block
  const int nt128 = NI == 2 ? nt : nt >> 1, sub0 = NI == 2 ? wn * 2 : (nt & 1) * 2 + wn;
  const u32x4* const bp = reinterpret_cast<const u32x4*>(Wx) + ((size_t)nt128 * (2 * nk) * 12 + sub0 * 3) * 64 + lane;
  u32x4 br[2][3 * NI];      // [buffer][n tile * 3 + piece] of a K = 16 half step, one half step ahead
  auto load_b = [&](int kk, auto buf_t) {
    constexpr int buf = decltype(buf_t)::value;
#pragma unroll
    for (int f = 0; f < 3 * NI; ++f) br[buf][f] = bp[((size_t)kk * 12 + f) * 64];
  };

  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int aoff = (wm * 64 + lr) * AP + lh * 8;   // this lane's fragment: row wm*64 + mi*32 + lr, k = 16 ss + 8 lh .. + 7
  auto half_step = [&](int buf, int ss, auto bb_t) {
    constexpr int bb = decltype(bb_t)::value;
    u32x4 pa[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        pa[mi][q] = *reinterpret_cast<const u32x4*>(As + buf * ABUF + (q * 128 + mi * 32) * AP + aoff + ss * 16);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x16 t = acc[mi][ni];
        // smallest terms first
        t = mfma_bf16(pa[mi][2], br[bb][ni * 3 + 0], t);
        t = mfma_bf16(pa[mi][0], br[bb][ni * 3 + 2], t);
        t = mfma_bf16(pa[mi][1], br[bb][ni * 3 + 1], t);
        t = mfma_bf16(pa[mi][1], br[bb][ni * 3 + 0], t);
        t = mfma_bf16(pa[mi][0], br[bb][ni * 3 + 1], t);
        t = mfma_bf16(pa[mi][0], br[bb][ni * 3 + 0], t);
        acc[mi][ni] = t;
      }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  const int nkk = 2 * nk;
  // prologue: step 0 in LDS, step 1 in register set 1, the first weight fragments on their way
  load_a(0, B0{});
  load_a(min(1, nk - 1), B1{});
  load_b(0, B0{});
  store_a(0, B0{});   // waits for set 0 only (loads retire in order)
  lds_barrier();
  // two steps per trip so that the register set of a load is a compile-time fact; every load and every LDS store is unconditional
  // (clamped index; the last step re-stores its own tile into the buffer nobody reads any more), so a step is ONE basic block and the
  // split of the next tile can be scheduled between the MFMAs of the second half step (a wave issues in order: 24 MFMAs back to back
  // stall it for 23 MFMA times while its ~100 VALU operations wait behind them)
  auto second_half_and_stage = [&](int buf, auto bb_t, auto set_t) {
    half_step(buf, 1, bb_t);
    store_a(buf ^ 1, set_t);
#pragma unroll
    for (int k = 0; k < 12 * NI; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NI == 2 ? 4 : 8, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x200, 12, 0);
  };
  for (int s = 0; s < nk; s += 2) {
    // ---- step s (LDS buffer 0); set 1 holds step s + 1, set 0 is free
    load_a(min(s + 2, nk - 1), B0{});
    load_b(min(2 * s + 1, nkk - 1), B1{});
    __builtin_amdgcn_sched_barrier(0);
    half_step(0, 0, B0{});
    load_b(min(2 * s + 2, nkk - 1), B0{});
    __builtin_amdgcn_sched_barrier(0);
    second_half_and_stage(0, B1{}, B1{});
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    if (s + 1 >= nk) break;
    // ---- step s + 1 (LDS buffer 1); set 0 holds step s + 2, set 1 is free
    load_a(min(s + 3, nk - 1), B1{});
    load_b(min(2 * s + 3, nkk - 1), B1{});
    __builtin_amdgcn_sched_barrier(0);
    half_step(1, 0, B0{});
    load_b(min(2 * s + 4, nkk - 1), B0{});
    __builtin_amdgcn_sched_barrier(0);
    second_half_and_stage(1, B1{}, B0{});
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); 32 lanes store the 128
  // contiguous bytes of 32 output channels of one output pixel
  float* const tile_out = MODE == 0 ? out + (size_t)pix0 * ldout + coff : out + (size_t)bm0 * ldout + coff;
  const bool full_m = bm0 + 128 <= M;
  auto store_tile = [&](auto guarded_t) {
    constexpr bool GUARDED = decltype(guarded_t)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = nt * (64 * NI) + wn * (32 * NI) + ni * 32 + lr;
      int ncol = n;
      if (MODE == 0) {
        const int q = n / Cout;
        ncol = ((q >> 1) * Wout + (q & 1)) * ldout + (n - q * Cout);
      }
      const float sh = shift ? shift[n] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rrow = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const unsigned idx = (unsigned)((MODE == 0 ? rowoff[rrow] : rrow * ldout) + ncol);
          const float v = acc[mi][ni][r] + sh;
          if (!GUARDED) tile_out[idx] = v;
          else if (bm0 + rrow < M) tile_out[idx] = v;
        }
    }
  };
#if defined(MGU_DIAG) && MGU_DIAG == 7   // diagnostic build: no output stores (accumulators kept live)
  {
    float sacc = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[mi][ni][r];
    if (sacc == 123.456f) tile_out[tid] = sacc;
  }
#else
  if (full_m) store_tile(std::false_type{});
  else store_tile(std::true_type{});
#endif
}

}  // namespace

size_t convt_x3_floats(int Cin, int Cout) { return (size_t)Cin * Cout * 6; }   // 4 Cout columns x Cin x 3 pieces x 2 bytes


namespace {

// Weights of the data gradient in the same fragment layout: k = q * Cout + co (q = qy*2 + qx), n = ci.  Columns past Cin inside the last
// 128-column block are never read (a 64-column workgroup tile reads its own half).
__global__ void pack_convt_x3_dgrad_kernel(const float* __restrict__ w, uint16_t* __restrict__ Wx, int Cin, int Cout) {
  pack_convt_x3_body(w, Wx, Cin, Cout, 1, blockIdx.x, gridDim.x);
}

}  // namespace

size_t convt_x3_dgrad_floats(int Cin, int Cout) { return (size_t)((Cin + 127) / 128 * 128) * Cout * 6; }   // x 4 taps x 3 pieces x 2 B / 4

hipError_t launch_pack_convt_x3_dgrad(const float* w, float* Wx, int Cin, int Cout, hipStream_t s) {
  if ((Cin & 63) || (Cout & 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_convt_x3_dgrad_kernel, dim3((unsigned)std::min<int64_t>(4096, ((int64_t)Cin * Cout * 4 + 255) / 256)), dim3(256), 0, s, w,
                     reinterpret_cast<uint16_t*>(Wx), Cin, Cout);
  return hipGetLastError();
}

// d: the KS = 2 gather descriptor of the generic path (in = dout + c_off with pitch ldin, Cp = the layer's Cout, N = the layer's Cin,
// H x W the input grid, Hout x Wout the grid of dout), d.wu = launch_pack_convt_x3_dgrad's panel
bool convt_x3_dgrad_applicable(const IgemmDesc& d) {
  return d.KS == 2 && d.wu && d.out_mode == 0 && (d.Cp & 31) == 0 && d.K == 4 * d.Cp && (d.N & 63) == 0 && (d.ldin & 3) == 0 && !d.scale &&
         !d.shift && !d.relu && !d.split_n && !d.pool && tun(d).wino_prec != 0 && (long)d.M * d.ldout < (1l << 31) &&
         (long)d.Hout * d.Wout * d.ldin < (1l << 31);
}

hipError_t launch_convt_x3_dgrad(const IgemmDesc& d, hipStream_t s) {
  const bool wide = (d.N & 127) == 0;
  const int mtiles = (d.M + 127) / 128, ntn = d.N / (wide ? 128 : 64);
  const int nb = mtiles * ntn;
  const int chunk = (nb + 7) / 8;
  if (wide)
    hipLaunchKernelGGL((convt2x2_x3_kernel<1, 2>), dim3(chunk * 8), dim3(256), 0, s, d.in, d.ldin, reinterpret_cast<const uint16_t*>(d.wu),
                       (const float*)nullptr, d.out, d.M, d.H, d.W, d.Cp, d.N, d.ldout, d.coff, d.Hout, d.Wout, ntn, nb);
  else
    hipLaunchKernelGGL((convt2x2_x3_kernel<1, 1>), dim3(chunk * 8), dim3(256), 0, s, d.in, d.ldin, reinterpret_cast<const uint16_t*>(d.wu),
                       (const float*)nullptr, d.out, d.M, d.H, d.W, d.Cp, d.N, d.ldout, d.coff, d.Hout, d.Wout, ntn, nb);
  return hipGetLastError();
}

hipError_t launch_pack_convt_x3(const float* w, float* Wx, int Cin, int Cout, hipStream_t s) {
  if ((Cin & 15) || (Cout & 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_convt_x3_kernel, dim3((unsigned)std::min<int64_t>(4096, ((int64_t)Cin * Cout * 4 + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<uint16_t*>(Wx), Cin,
                     Cout);
  return hipGetLastError();
}

bool convt_x3_applicable(const IgemmDesc& d) {
  return d.out_mode == 1 && d.wu && d.KS == 1 && d.K == d.Cp && (d.Cp & 31) == 0 && (d.ct_cout & 31) == 0 && d.N == 4 * d.ct_cout &&
         (d.ldin & 3) == 0 && !d.scale && !d.relu && !d.split_n && tun(d).wino_prec != 0 &&
         (9l * 128 + 8l * d.Wout) * d.ldout < (1l << 31);   // the output pixels of a tile's 128 rows span < 2^31 elements
}

hipError_t launch_convt_x3(const IgemmDesc& d, hipStream_t s) {
  const int mtiles = (d.M + 127) / 128, ntn = d.N / 128;
  const int nb = mtiles * ntn;
  const int chunk = (nb + 7) / 8;
  hipLaunchKernelGGL((convt2x2_x3_kernel<0, 2>), dim3(chunk * 8), dim3(256), 0, s, d.in, d.ldin, reinterpret_cast<const uint16_t*>(d.wu), d.shift, d.out,
                     d.M, d.H, d.W, d.Cp, d.ct_cout, d.ldout, d.coff, d.Hout, d.Wout, ntn, nb);
  return hipGetLastError();
}

}  // namespace mgu
