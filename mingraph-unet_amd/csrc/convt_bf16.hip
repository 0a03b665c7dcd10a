// ConvTranspose2d(kernel 2, stride 2) of the bf16-storage mode (model/unet/unet_decoder.py:25,36; BASELINE configs[2]) on the bf16
// matrix cores: the bf16 sibling of convt2x2_x3_kernel (convt_x3.hip) -- one operand piece instead of three, the same GEMM
//   out[(2y+dy, 2x+dx)][co] = sum_ci in[(y,x)][ci] * w[ci][co][dy][dx]      (M = B H W pixels, N = 4 Cout, K = Cin).
// The generic tile kernel it replaces (igemm_kernel<bf16,1,1>) ran the four layers at 9 % of the bf16 peak: its epilogue writes one
// 2-byte value per lane -- 64-byte half lines -- and its K loop is the generic gather.  (A first sibling that fetched MFMA fragments
// straight from global memory, 32-byte row segments per K slice, had measured slower than that kernel: mgunet_api.hip.)  Here:
//   * A: the 128 x 64 bf16 tile of a K = 64 step is read ONCE per workgroup in whole 128-byte rows (8 lanes x 16 B per row), one step
//     ahead in registers, and parked in a double-buffered LDS tile with 144-byte rows (conflict-free ds_read_b128 fragments);
//   * B: weights converted and laid out per lane at load time (pack_convt_bf16f_kernel): a wave's fragment of a K = 16 half step is one
//     16-byte load from a block every workgroup of the same n tile reads (L2 resident), one half step ahead;
//   * epilogue: bias add in fp32, conversion to bf16, then a wave-private LDS transpose so that every lane stores 16 BYTES (eight
//     consecutive output channels of one output pixel): whole 64- or 128-byte runs per pixel instead of 2-byte lanes;
//   * XCD-aware workgroup order (the n tiles of a pixel tile share an L2), as in the fp32 kernel.
// Workgroup = 4 waves on 128 pixels x 128 columns, wave tile 64 x 64 (2 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace mgu {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// LDS hand-off barrier without the vmcnt(0) of __syncthreads()'s fence (the prefetched loads stay in flight); the empty asm statements
// are compiler-only ordering points for the LDS accesses on both sides
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Wf[n / 128][k / 16][(n / 32) & 3][lane = 32 * ((k / 8) & 1) + (n & 31)][k & 7] = bf16(w[ci = k][co][dy][dx]),  n = (dy*2+dx)*Cout + co:
// the B fragment of v_mfma_f32_32x32x16_bf16, one 16-byte lane load
__global__ void pack_convt_bf16f_kernel(const float* __restrict__ w, __bf16* __restrict__ Wf, int Cin, int Cout) {
  const int64_t total = (int64_t)Cin * Cout * 4;
  const int ksteps = Cin >> 4;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Cin), n = (int)(idx / Cin);
    const int q = n / Cout, co = n - q * Cout;
    const float x = w[(((int64_t)k * Cout + co) * 2 + (q >> 1)) * 2 + (q & 1)];
    const int lane = ((k >> 3) & 1) * 32 + (n & 31);
    Wf[(((((int64_t)(n >> 7) * ksteps + (k >> 4)) * 4 + ((n >> 5) & 3))) * 64 + lane) * 8 + (k & 7)] = (__bf16)x;
  }
}

__global__ __launch_bounds__(256, 2) void convt2x2_bf16_kernel(const __bf16* __restrict__ in, const int ldin, const __bf16* __restrict__ Wf,
                                                               const float* __restrict__ shift, __bf16* __restrict__ out, const int M, const int H,
                                                               const int W, const int Cin, const int Cout, const int ldout, const int coff,
                                                               const int Hout, const int Wout, const int ntn, const int nblocks) {
  constexpr int AP = 72, ABUF = 128 * AP;               // A tile of a K = 64 step: 128 rows x (64 + 8 pad) bf16 (144-byte rows)
  __shared__ __attribute__((aligned(16))) uint16_t As[2 * ABUF];   // 36 864 B; the epilogue's transpose tiles (4 x 4 608 B) reuse it
  __shared__ int rowoff[128];
  const int chunk = gridDim.x >> 3;
  const int lb = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);     // XCD-aware order (convt_x3.hip)
  if (lb >= nblocks) return;
  const int nt = lb % ntn, mt = lb / ntn;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int bm0 = mt * 128;
  const int HW = H * W;
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
      off = (int)((((long long)img * Hout + 2 * y) * Wout + 2 * x - pix0) * ldout);
    }
    rowoff[tid] = off;
  }
  // ---- A staging: thread (row arow + 32 j, 16-byte piece ach) of the 128 x 64 tile
  const int arow = tid >> 3, ach = (tid & 7) * 8;
  const __bf16* aptr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) aptr[j] = in + (size_t)min(bm0 + j * 32 + arow, M - 1) * ldin + ach;   // rows past the end: re-read, never stored
  const int nk = Cin >> 6;   // K = 64 steps
  u32x4 areg[4];
  auto load_a = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) areg[j] = *reinterpret_cast<const u32x4*>(aptr[j] + s * 64);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(As + buf * ABUF + (j * 32 + arow) * AP + ach) = areg[j];
  };
  // ---- B fragments: [n / 128][half step][column tile][lane] x 16 B; this wave's two column tiles are 2 wn, 2 wn + 1
  const u32x4* const bp = reinterpret_cast<const u32x4*>(Wf) + ((size_t)nt * (4 * nk) * 4 + wn * 2) * 64 + lane;
  u32x4 br[2][2];
  auto load_b = [&](int kk, int buf) {
#pragma unroll
    for (int f = 0; f < 2; ++f) br[buf][f] = bp[((size_t)kk * 4 + f) * 64];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int aoff = (wm * 64 + lr) * AP + lh * 8;
  const int nkk = 4 * nk;
  load_a(0);
  load_b(0, 0);
  store_a(0);
  if (nk > 1) load_a(1);
  lds_barrier();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int kk = 4 * s + ss;
      load_b(min(kk + 1, nkk - 1), (ss + 1) & 1);     // (4 half steps per step: the buffer parity is static)
      u32x4 pa[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) pa[mi] = *reinterpret_cast<const u32x4*>(As + buf * ABUF + mi * 32 * AP + aoff + ss * 16);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_bf16(pa[mi], br[ss & 1][ni], acc[mi][ni]);
    }
    if (s + 1 < nk) {
      store_a(buf ^ 1);                               // step s + 1 (in registers since the previous step) -> the idle buffer
      if (s + 2 < nk) load_a(s + 2);
    }
    lds_barrier();
  }
  // ---- epilogue: bias, bf16, wave-private transpose, 16-byte stores.  (The barrier above: every wave has left the A tiles.)
  __bf16* const tile_out = out + (size_t)pix0 * ldout + coff;
  uint16_t* const Ts = As + wave * (32 * AP);          // 32 rows x 64 columns (+ pad) of this wave's current m tile
  const int ncol0 = nt * 128 + wn * 64;                 // first column of the wave
  float sh[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) sh[ni] = shift ? shift[ncol0 + ni * 32 + lr] : 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        Ts[row * AP + ni * 32 + lr] = __builtin_bit_cast(uint16_t, (__bf16)(acc[mi][ni][r] + sh[ni]));
      }
    // the wave's own LDS writes are complete and visible to its own lanes (LDS operations of a wave execute in order; the compiler
    // must not move the reads above the writes)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = it * 64 + lane, row = id >> 3, g = id & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Ts + row * AP + g * 8);
      const int n = ncol0 + g * 8, q = n / Cout, co = n - q * Cout;
      const int rrow = wm * 64 + mi * 32 + row;
      if (bm0 + rrow < M)
        *reinterpret_cast<u32x4*>(tile_out + rowoff[rrow] + ((q >> 1) * Wout + (q & 1)) * ldout + co) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                    // the next m tile overwrites the transpose tile
  }
}

}  // namespace

size_t convt_bf16f_floats(int Cin, int Cout) { return (size_t)Cin * Cout * 2; }   // 4 Cout columns x Cin x 2 bytes

hipError_t launch_pack_convt_bf16f(const float* w, float* Wf, int Cin, int Cout, hipStream_t s) {
  if ((Cin & 63) || (Cout & 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_convt_bf16f_kernel, dim3((unsigned)std::min<int64_t>(4096, ((int64_t)Cin * Cout * 4 + 255) / 256)), dim3(256), 0, s, w,
                     reinterpret_cast<__bf16*>(Wf), Cin, Cout);
  return hipGetLastError();
}

// d: the ConvTranspose descriptor of the bf16 mode (in / out point to bf16; d.wu = launch_pack_convt_bf16f's fragments)
bool convt_bf16f_applicable(const IgemmDesc& d) {
  return d.out_mode == 1 && d.wu && d.KS == 1 && d.K == d.Cp && (d.Cp & 63) == 0 && (d.ct_cout & 31) == 0 && d.N == 4 * d.ct_cout &&
         (d.ldin & 7) == 0 && (d.ldout & 7) == 0 && (d.coff & 7) == 0 && !d.scale && !d.relu && !d.split_n && tun(d).convt_frag &&
         (9l * 128 + 8l * d.Wout) * d.ldout < (1l << 31);   // the output pixels of a tile's 128 rows span < 2^31 elements
}

hipError_t launch_convt_bf16f(const IgemmDesc& d, hipStream_t s) {
  const int mtiles = (d.M + 127) / 128, ntn = d.N / 128;
  const int nb = mtiles * ntn;
  const int chunk = (nb + 7) / 8;
  hipLaunchKernelGGL(convt2x2_bf16_kernel, dim3(chunk * 8), dim3(256), 0, s, reinterpret_cast<const __bf16*>(d.in), d.ldin,
                     reinterpret_cast<const __bf16*>(d.wu), d.shift, reinterpret_cast<__bf16*>(d.out), d.M, d.H, d.W, d.Cp, d.ct_cout, d.ldout,
                     d.coff, d.Hout, d.Wout, ntn, nb);
  return hipGetLastError();
}

}  // namespace mgu
