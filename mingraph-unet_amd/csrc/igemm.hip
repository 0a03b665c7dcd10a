// Implicit-GEMM convolution / transposed-convolution / linear kernels for gfx950 (CDNA4): exact fp32, and bf16
// storage with fp32 accumulation (one source, templated on the element type).
//
// Replaces the aten::mkldnn_convolution calls behind ConvBlock (model/unet/unet_encoder.py:4-25),
// ConvTranspose2d (model/unet/unet_decoder.py:25,36), final_conv (unet_decoder.py:117,143) and the
// nn.Linear W of GraphAttentionLayer (model/gat/graph_attention.py:28,53).
//
// Design (MI355X-first, not a translation of anything):
//   * exact-fp32 matrix cores: v_mfma_f32_32x32x2_f32, 64-lane wavefronts, accumulators in VGPR/AGPR;
//     this is the fp32 roofline of the chip (157 TFLOP/s), 1/16 of the bf16 rate, so every conv of the
//     fp32 configuration is MFMA-bound, not HBM-bound (arithmetic intensity >= 72 F/B vs balance ~25);
//   * NHWC activations: a pixel's channels are contiguous, so the im2col gather of a 3x3 tap is a
//     16-byte-per-lane coalesced global load (8 lanes cover one 128-byte line of a pixel);
//   * LDS tiles As[BM][36], Bs[BN][36] (k contiguous, +4 floats pad): the MFMA operand reads are
//     conflict-free ds_read_b128 (row pitch 36 dwords spreads any 16 rows over all 64 banks);
//   * register-staged software pipeline: the global loads of K-step s+1 are issued before the MFMA
//     block of step s and parked in VGPRs, then written to the (single) LDS buffer after it;
//     2-3 workgroups per CU overlap each other's barrier/LDS-write bubbles;
//   * fused epilogue: y = relu(scale[n]*acc + shift[n]) (conv bias + BatchNorm folded), stored with a
//     channel pitch/offset so encoder outputs and the pixel-shuffled ConvTranspose outputs land
//     directly in the two halves of the decoder's concat buffer (torch.cat never materialises).
#include "common.h"
#include "x3.h"
#include <algorithm>
#include <type_traits>

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

const Tuning& default_tuning() {
  static const Tuning t;
  return t;
}

hipError_t ensure_dyn_lds(const void* func, size_t bytes, bool (&done)[64]) {
  if (bytes <= 65536) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done[dev] = true;
  return e;
}

// ---- element traits: the kernels are written once for fp32 (exact v_mfma_f32_32x32x2_f32) and bf16 storage with
// fp32 accumulation (v_mfma_f32_32x32x16_bf16).  All staging moves raw 16-byte chunks (4 floats or 8 bf16); an LDS
// row holds NP such chunks of a pixel / output channel plus one chunk of padding (pitch 144 B for NP = 8, 80 B for
// NP = 4: both map any 16 rows of a ds_read_b128 lane group to 16 distinct 4-bank slots).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int VEC = 4; };
template <> struct Elem<__bf16> { static constexpr int VEC = 8; };

// acc += A(32 x kchunk) * B(kchunk x 32) for the 16-byte operand chunks of one lane: fp32 = 4 MFMAs of K = 2 (lane
// (r, h) holds k = 4h .. 4h+3, element t feeds MFMA t), bf16 = one MFMA of K = 16 (lane holds k = 8h .. 8h+7)
template <typename T>
__device__ __forceinline__ f32x16 mma_chunk(const f32x4 a, const f32x4 b, f32x16 c) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], c, 0, 0, 0);
    return c;
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

template <typename T, int KS, int OUTMODE, int WAVES_M, int WAVES_N, int WMT, int WNT>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmDesc d) {
  static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per workgroup");
  constexpr int VEC = Elem<T>::VEC;      // elements per 16-byte chunk
  constexpr int CK = 8 * VEC;            // K elements per pipeline step (one 128-byte row)
  constexpr int LDS_LD = CK + VEC;       // LDS row pitch in elements (144 B)
  constexpr int BM = WAVES_M * WMT * 32;
  constexpr int BN = WAVES_N * WNT * 32;
  constexpr int AR = BM / 32;  // A rows staged per thread
  constexpr int BR = BN / 32;  // B rows staged per thread
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  T* As = smem;
  T* Bs = smem + BM * LDS_LD;
  const T* const in_t = reinterpret_cast<const T*>(d.in);
  const T* const w_t = reinterpret_cast<const T*>(d.w);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int bm0 = blockIdx.x * BM;
  const int bn0 = blockIdx.y * BN;
  const int kq = tid & 7;    // which 16-byte chunk of the K slice this thread stages
  const int r0 = tid >> 3;   // first tile row this thread stages (then +32, +64, ...)

  // ---- per-row gather state: pixel base pointer + 9-bit tap validity mask ---------------------
  const T* abase[AR];
  unsigned amask[AR];
  const int HW = d.H * d.W;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = bm0 + r0 + 32 * i;
    abase[i] = in_t;
    amask[i] = 0u;
    if (m < d.M) {
      abase[i] = in_t + (size_t)m * d.ldin;
      if (KS == 2) {
        // stride-2 2x2 gather (ConvTranspose2d dgrad): row m = (img, y, x) over H x W reads the four
        // pixels (2y+dy, 2x+dx) of the Hout x Wout source grid; every tap is in range (Hout >= 2H).
        const int img = m / HW;
        const int rem = m - img * HW;
        const int oy = rem / d.W;
        const int ox = rem - oy * d.W;
        abase[i] = in_t + (((size_t)img * d.Hout + 2 * oy) * d.Wout + 2 * ox) * d.ldin;
        amask[i] = 0xFu;
      } else if (KS == 3) {
        const int rem = m % HW;
        const int oy = rem / d.W;
        const int ox = rem - oy * d.W;
        unsigned rowok = (oy > 0 ? 1u : 0u) | 2u | (oy + 1 < d.H ? 4u : 0u);
        unsigned colok = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < d.W ? 4u : 0u);
        unsigned mk = 0u;
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (rowok & (1u << r)) mk |= colok << (3 * r);
        amask[i] = mk;
      } else {
        amask[i] = 1u;
      }
    }
  }
  const T* wrow[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) wrow[i] = w_t + (size_t)(bn0 + r0 + 32 * i) * d.Kp + kq * VEC;

  f32x4 areg[AR];
  f32x4 breg[BR];
  auto load_tiles = [&](int k0) {
    const int k = k0 + kq * VEC;
    int tap = 0;
    long delta = k;
    if (KS == 3) {
      tap = k / d.Cp;
      const int c = k - tap * d.Cp;
      const int r = tap / 3;
      const int s = tap - 3 * r;
      delta = (long)((r - 1) * d.W + (s - 1)) * d.ldin + c;
    } else if (KS == 2) {
      tap = k / d.Cp;
      const int c = k - tap * d.Cp;
      delta = (long)((tap >> 1) * d.Wout + (tap & 1)) * d.ldin + c;
    }
    const bool kvalid = k < d.K;
    // NO branch around the gathers: a conditional load makes hipcc drain vmcnt per element (the loads of a K-step
    // would serialise).  Invalid taps read d.in[0..3] (always mapped) and are zeroed by a select.
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = kvalid && ((amask[i] >> tap) & 1u);
      const T* src = ok ? abase[i] + delta : in_t;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      areg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) breg[i] = *reinterpret_cast<const f32x4*>(wrow[i] + k0);
  };

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int lr = lane & 31;
  const int lh = lane >> 5;
  const T* Ap = As + (wm * WMT * 32 + lr) * LDS_LD + lh * VEC;
  const T* Bp = Bs + (wn * WNT * 32 + lr) * LDS_LD + lh * VEC;
  T* Asw = As + r0 * LDS_LD + kq * VEC;
  T* Bsw = Bs + r0 * LDS_LD + kq * VEC;

  const int nk = d.Kp / CK;
  load_tiles(0);
  for (int ks = 0; ks < nk; ++ks) {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(Asw + i * 32 * LDS_LD) = areg[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(Bsw + i * 32 * LDS_LD) = breg[i];
    __syncthreads();
    if (ks + 1 < nk) load_tiles((ks + 1) * CK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (ks * CK + kk * 2 * VEC >= d.K) break;  // zero K tail (block-uniform): nothing to accumulate
      f32x4 a[WMT], b[WNT];
#pragma unroll
      for (int mi = 0; mi < WMT; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(Ap + mi * 32 * LDS_LD + kk * 2 * VEC);
#pragma unroll
      for (int ni = 0; ni < WNT; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Bp + ni * 32 * LDS_LD + kk * 2 * VEC);
      // lane (lr, lh) holds the 16-byte chunk k = kk*2*VEC + lh*VEC .. +VEC-1 of its row; A and B use the same
      // (lh, element) -> k map, so the permuted k order inside a 2*VEC group is consistent.
#pragma unroll
      for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
        for (int ni = 0; ni < WNT; ++ni) acc[mi][ni] = mma_chunk<T>(a[mi], b[ni], acc[mi][ni]);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  // ConvTranspose (OUTMODE 1): the (img, y, x) decode of a row costs two integer divisions; doing it per
  // accumulator register made the epilogue ~5000 VALU instructions per wave (the K = 64 full-resolution layer
  // spent more time there than in its MFMAs).  Each row's output pixel index is computed ONCE per workgroup
  // into LDS (the tile buffers are free after the last barrier) and read back per register.
  // Every row's element offset (relative to the tile's first output pixel, 32 bit) is computed ONCE per workgroup into
  // LDS (the tile buffers are free after the last barrier) and read back per register.
  int* rowoff = reinterpret_cast<int*>(smem_raw);
  long long pix0 = 0;   // OUTMODE 1: output pixel of the tile's first row (block-uniform)
  if (OUTMODE == 1) {
    {
      const int img = bm0 / HW;
      const int rem = bm0 - img * HW;
      const int y = rem / d.W;
      pix0 = ((long long)img * d.Hout + 2 * y) * d.Wout + 2 * (rem - y * d.W);
    }
    for (int rrow = tid; rrow < BM; rrow += 256) {
      const int m = bm0 + rrow;
      int off = 0;
      if (m < d.M) {
        const int img = m / HW;
        const int rem = m - img * HW;
        const int y = rem / d.W;
        const int x = rem - y * d.W;
        off = (int)((((long long)img * d.Hout + 2 * y) * d.Wout + 2 * x - pix0) * d.ldout);   // a tile spans < 2^31 elements
      }
      rowoff[rrow] = off;
    }
    __syncthreads();
  }
  // Address arithmetic is kept off the per-element path: a uniform 64-bit tile base + a 32-bit element index, and
  // the stores of a full tile carry NO per-element condition (hipcc puts an s_waitcnt vmcnt(0) in front of every
  // conditional store, which serialises the store round trips: that, not the MFMAs, bounded the K-short layers).
  const bool full_m = bm0 + BM <= d.M;   // block-uniform
  const bool full_n = bn0 + BN <= d.N;   // block-uniform
  T* const out_t = reinterpret_cast<T*>(d.out);
  T* const tile_out = OUTMODE == 1 ? out_t + (size_t)pix0 * d.ldout + d.coff : out_t + (size_t)bm0 * d.ldout + d.coff;
  auto store_tile = [&](auto guarded_t) {
    constexpr bool GUARDED = decltype(guarded_t)::value;
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni) {
      const int n = bn0 + (wn * WNT + ni) * 32 + lr;
      const bool nvalid = n < d.N;
      const float sc = (nvalid && d.scale) ? d.scale[n] : 1.f;
      const float sh = (nvalid && d.shift) ? d.shift[n] : 0.f;
      int ncol = n;   // element offset of column n inside a row's pixel (OUTMODE 1: tap q = (dy, dx) + channel)
      if (OUTMODE == 1) {
        const int q = n / d.ct_cout;
        ncol = ((q >> 1) * d.Wout + (q & 1)) * d.ldout + (n - q * d.ct_cout);
      }
      const bool split = OUTMODE == 0 && d.split_n > 0 && n >= d.split_n;
#pragma unroll
      for (int mi = 0; mi < WMT; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int rrow = (wm * WMT + mi) * 32 + row;
          float v = acc[mi][ni][r] * sc + sh;
          if (d.relu) v = fmaxf(v, 0.f);
          const unsigned idx = OUTMODE == 1 ? (unsigned)(rowoff[rrow] + ncol) : (unsigned)(rrow * d.ldout + ncol);
          if (!GUARDED) {
            tile_out[idx] = (T)v;
          } else if (nvalid && (full_m || bm0 + rrow < d.M)) {
            if (split) d.out2[(size_t)(bm0 + rrow) * d.ld2 + (n - d.split_n)] = v;   // always fp32 (GAT scalars)
            else tile_out[idx] = (T)v;
          }
        }
      }
    }
  };
  if (full_m && full_n && !(OUTMODE == 0 && d.split_n > 0)) store_tile(std::false_type{});
  else store_tile(std::true_type{});
}

// =================================================================================================
// conv3x3 with an LDS-resident input halo (Cin % 32 == 0).
//
// The generic kernel above re-gathers the A tile from L2/HBM once per tap: 9x the input bytes, and at
// full resolution those re-reads miss the 4 MiB XCD L2 (rocprofv3 r01: dec3.conv1 fetched 4.0 GB for a
// 0.54 GB input, L2 hit 28 %).  Here a workgroup owns a TH x 16 pixel patch of one image: per 32-channel
// chunk it stages the (TH+2) x 18 halo ONCE into LDS and all 9 taps read their shifted A operands from
// it (a tap is just a constant LDS address offset).  Only the small per-tap weight tile streams through a
// double-buffered LDS panel, so there is a single s_barrier per tap.  The halo of the next chunk is
// prefetched into registers while the current chunk computes.
// =================================================================================================
#if defined(MGU_DIAG) && MGU_DIAG == 23   // diagnostic build: step timeline of one workgroup of the selected layer (tools/diag_timeline.py --halo)
#ifndef MGU_DIAG_H
#define MGU_DIAG_H 128
#endif
#ifndef MGU_DIAG_CP
#define MGU_DIAG_CP 128
#endif
#ifndef MGU_DIAG_N
#define MGU_DIAG_N 128
#endif
__device__ unsigned long long mgu_halo_ts[4][256][4];
#define HALO_T(slot)                                                                                                   \
  do {                                                                                                                 \
    if (diag_on && (threadIdx.x & 63) == 0 && st < 256) mgu_halo_ts[threadIdx.x >> 6][st][slot] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define HALO_T(slot) do {} while (0)
#endif
// max(lo, x) as ONE v_max_f32 (fmaxf() is two: the backend first quiets a possible signalling NaN).  A NaN operand yields the OTHER
// operand (IEEE mode), a NaN only if both are: callers that must pass x through untouched hand in lo = quiet NaN.
__device__ __forceinline__ float max_1op(const float lo, const float x) {
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(x));
  return r;
}

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x70 | 0xF00);
}

#if defined(MGU_DIAG) && MGU_DIAG == 50
__device__ unsigned mgu_diag_glds_bad;
__device__ unsigned long long mgu_diag_glds_n;
#endif
template <typename T, int NP, int TH, int WAVES_M, int WAVES_N, int WMT, int WNT, int TPS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const IgemmDesc d, const int tiles_x, const int tiles_y,
                                                           const int total_patches, const int patches_per_block, const int yfast) {
  constexpr int VEC = Elem<T>::VEC;      // elements per 16-byte chunk
  constexpr int CK = NP * VEC;           // channels per chunk (NP 16-byte pieces per pixel)
  constexpr int LDS_LD = CK + VEC;       // LDS row pitch in elements
  constexpr int RPP = 256 / NP;          // rows staged per pass of the 256 threads
  // A workgroup walks `patches_per_block` consecutive TH x 16 patches (x fastest, then y, then image).  The
  // unit of the pipeline is an ITEM = (patch, 32-channel chunk) = 9/TPS steps of TPS taps each (one s_barrier per
  // step: TPS = 3 gives the narrow N <= 32 tile 96 MFMAs per wave between barriers instead of 32); the halo of
  // item i+1 -- which may belong to the next patch -- is prefetched into registers during item i, so only the
  // very first halo load of a workgroup is exposed and a patch's epilogue stores overlap the next patch's loads.
  static_assert(TPS == 1 || TPS == 3, "taps per step");
  constexpr int SPI = 9 / TPS;  // steps per item
  constexpr int TW = 16, HWID = TW + 2, HP = (TH + 2) * HWID;
  constexpr int BM = TH * TW;
  constexpr int BN = WAVES_N * WNT * 32;
  static_assert(BM == WAVES_M * WMT * 32 && WAVES_M * WAVES_N == 4, "tile/wave mismatch");
  constexpr int HR = (HP + RPP - 1) / RPP;  // halo pixels staged per thread
  constexpr int BR = (BN + RPP - 1) / RPP;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  T* Hs = reinterpret_cast<T*>(smem_raw);   // [HP][LDS_LD]
  T* Bs = Hs + HP * LDS_LD;                 // [2][TPS][BN][LDS_LD]   (GLDS: [2][TPS][BN] rows of 128 bytes, XOR-swizzled, no pad)
  // GLDS (64-channel-chunk tiles): the weight tile of a later step goes global -> LDS by LDS-DMA
  // (global_load_lds_dwordx4: no registers, no ds_write pass, no wait in front of a hand-off).  An instruction writes 1 KB = eight
  // 128-byte rows in lane order, so the rows cannot be padded: 16-byte piece q of row r sits at position q ^ ((r >> 1) & 7) (the
  // SOURCE address is per lane, the image is linear) and the fragment reads of 16 consecutive rows fall on 16 distinct bank groups.
#if defined(MGU_HALO_NO_GLDS)   // (A/B build: weight tiles through registers, as the 32-channel-chunk tiles do)
  constexpr bool GLDS = false;
#else
  constexpr bool GLDS = NP == 8 && BN % 32 == 0;
#endif
  constexpr int BROW = GLDS ? CK : LDS_LD;   // weight-row pitch in LDS (elements)
  // GLDS: three weight buffers (the tile of step s + 2 requested at the top of step s) where two workgroups of the CU still fit
  constexpr int NBUF = (GLDS && (HP * LDS_LD + 3 * TPS * BN * CK) * (int)sizeof(T) <= 80 * 1024) ? 3 : 2;
  constexpr int NGL = TPS * (BN / 32);       // (GLDS) DMA instructions per wave and step
  const T* const in_t = reinterpret_cast<const T*>(d.in);
  const T* const w_t = reinterpret_cast<const T*>(d.w);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int kq = tid % NP;
  const int r0 = tid / NP;
  const int bn0 = blockIdx.y * BN;
  const int p_begin = blockIdx.x * patches_per_block;
  const int npatch = min(patches_per_block, total_patches - p_begin);
  if (npatch <= 0) return;

  // halo staging map of the patch being LOADED: thread -> (halo pixel r0 + 32 i, 16-byte piece kq), as BYTE offsets into a buffer
  // descriptor of the patch's image.  An out-of-image pixel gets an offset past the descriptor's range: the load returns zeros by
  // itself -- no mask, no select when the registers go to LDS (44 v_cndmask per item of the 64-channel-chunk tiles before).
  int hoff[HR];
  auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in_t), 0, 0x7ffffff0, 0x00020000);
  const unsigned img_bytes = (unsigned)((size_t)d.H * d.W * d.ldin * sizeof(T));
  // x fastest by default; y fastest (MGU_WINO_YFAST=1, as in wino3x3_cp_kernel) measured neutral (+-0.5 %) in the bf16 mode
  struct PatchPos { int img, y0, x0; };
  auto setup_patch = [&](int p) {   // (by value: out-parameters through the nested closures of the item loop ended up in scratch)
    const int ty = yfast ? p % tiles_y : (p / tiles_x) % tiles_y;
    const int tx = yfast ? (p / tiles_y) % tiles_x : p % tiles_x;
    return PatchPos{p / (tiles_x * tiles_y), ty * TH, tx * TW};
  };
  auto setup_load = [&](int p) {
    const PatchPos pp = setup_patch(p);
    const int img = pp.img, y0 = pp.y0, x0 = pp.x0;
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in_t) + (size_t)img * d.H * d.W * d.ldin, 0, img_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = r0 + RPP * i;
      const int hy = hp / HWID, hx = hp - hy * HWID;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = hp < HP && y >= 0 && y < d.H && x >= 0 && x < d.W;
      hoff[i] = ok ? ((y * d.W + x) * d.ldin + kq * VEC) * (int)sizeof(T) : 0x7fff0000;
    }
  };
  const T* wrow[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) wrow[i] = w_t + (size_t)(bn0 + r0 + RPP * i) * d.Kp + kq * VEC;   // panel rows padded to 128

  // DEEPH (the 32-channel-chunk tiles: the full-resolution layers, one or two items per patch): the halo is requested TWO items ahead
  // into a second register set.  An item of those tiles is three short steps, less than an HBM round trip under load, and the
  // layers ran at 3.2 TB/s with neither pipe busy (the wino3x3_cp_kernel DEEP case).  The 64-channel tiles have no registers for it.
  constexpr bool DEEPH = NP == 4 && HR <= 6;
  constexpr int NSETH = DEEPH ? 2 : 1;
  f32x4 hreg[NSETH][HR];
  f32x4 breg[TPS][BR];
  auto load_halo = [&](int c, auto set_c) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < HR; ++i)
      hreg[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, hoff[i], c * CK * (int)sizeof(T), 0));
  };
  auto store_halo = [&](auto set_c) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < HR; ++i)
      if (r0 + RPP * i < HP) *reinterpret_cast<f32x4*>(Hs + (r0 + RPP * i) * LDS_LD + kq * VEC) = hreg[set][i];
  };
  auto load_b = [&](int c, int stp) {   // the TPS weight tiles of step `stp` of chunk c
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int k0 = (stp * TPS + tt) * d.Cp + c * CK;
#pragma unroll
      for (int i = 0; i < BR; ++i) breg[tt][i] = *reinterpret_cast<const f32x4*>(wrow[i] + k0);
    }
  };
  // GLDS: the TPS weight tiles of step `stp` of chunk c straight into buffer `buf`: per tap BN / 8 instructions of 1 KB, BN / 32 per wave
  auto glds_b = [&](int c, int stp, int buf) {
    if constexpr (GLDS) {
      const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int k0 = (stp * TPS + tt) * d.Cp + c * CK;
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) {
          const int ii = wave_u * (BN / 32) + j;                 // instruction = rows 8 ii .. 8 ii + 7 of the tile
          const int row = 8 * ii + (lane >> 3);
          const int logical = (lane & 7) ^ ((row >> 1) & 7);     // the piece that belongs at this lane's LDS position
          const T* src = w_t + (size_t)(bn0 + row) * d.Kp + k0 + logical * VEC;
          T* dst = Bs + ((size_t)(buf * TPS + tt) * BN + 8 * ii) * BROW;   // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of hipcc has no target feature for this builtin and silently drops the kernel's stub)
          __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
          (void)src, (void)dst;
#endif
        }
      }
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
      for (int i = 0; i < BR; ++i)
        if (BN % RPP == 0 || r0 + RPP * i < BN)
          *reinterpret_cast<f32x4*>(Bs + ((buf * TPS + tt) * BN + r0 + RPP * i) * LDS_LD + kq * VEC) = breg[tt][i];
  };

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int lr = lane & 31;
  const int lh = lane >> 5;
  int aoff[WMT];  // float offset of this lane's A row (tap 0,0) inside Hs
#pragma unroll
  for (int mi = 0; mi < WMT; ++mi) {
    const int pp = (wm * WMT + mi) * 32 + lr;
    aoff[mi] = ((pp >> 4) * HWID + (pp & 15)) * LDS_LD + lh * VEC;
  }
  const int boff = GLDS ? (wn * WNT * 32 + lr) * BROW : (wn * WNT * 32 + lr) * LDS_LD + lh * VEC;
  const int bswz = ((wn * WNT * 32 + lr) >> 1) & 7;   // (GLDS) this lane's row swizzle: the same for its WNT rows (32 apart)

  const int nchunks = d.Cp / CK;
  const int nitems = npatch * nchunks;
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, DEEPH ? 1 : 0>;
  // loader position: (patch, chunk) of the next item whose halo is requested; past the last item the last one is requested again
  int li = 0;
  auto load_next_halo = [&](auto set_c) {
    const int lic = min(li, nitems - 1);
    const int lp = lic / nchunks, lc = lic - lp * nchunks;
    if (lc == 0 && li < nitems) setup_load(p_begin + lp);   // a new patch (address arithmetic only; a clamped repeat keeps hoff)
    load_halo(lc, set_c);
    ++li;
  };
  // workgroup barrier.  GLDS: the raw form -- __syncthreads() drains vmcnt while an LDS-DMA is in flight (its fence), i.e. the weight
  // tile requested at the top of the step would be waited for at once; the waits for the DMA are counted by hand below
  // The raw form carries no fence of its own (s_barrier is IntrNoMem), so the two compiler-only ordering points keep every LDS access
  // of the source on its side of the barrier -- the fragment reads of Bs / Hs must not rise above it, the halo stores must not sink
  // below it -- without emitting a wait: an empty asm with a "memory" clobber generates no instruction and, unlike a workgroup fence,
  // no vmcnt(0).
  auto wg_barrier = [&]() {
    if constexpr (GLDS) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      __syncthreads();
    }
  };
  // The counted vmcnt waits below name LOWER bounds of what this wave has in flight behind the awaited DMA (a smaller count waits
  // for more, never for less): an interior patch's epilogue issues exactly one buffer store per accumulator register (store_patch,
  // !GUARDED: WMT * WNT * 16; the pooled stores come on top) and load_halo issues exactly HR buffer loads -- the two constants the
  // waits are built from.  vmcnt counts to 63.
  constexpr int STORES_PER_INTERIOR_PATCH = WMT * WNT * 16;
  constexpr int NST = STORES_PER_INTERIOR_PATCH < 63 ? STORES_PER_INTERIOR_PATCH : 63;
  static_assert(NST <= STORES_PER_INTERIOR_PATCH && NST <= 63, "vmcnt count above the stores really issued");
  int w0 = 0;                // (GLDS) stores issued behind the weight DMA of the NEXT item's first step: NST after an interior patch
  load_next_halo(Set0{});
  if constexpr (GLDS) {
    glds_b(0, 0, 0);
    if constexpr (NBUF == 3) glds_b(0, 1, 1);
    store_halo(Set0{});
    wait_vmcnt<0>();
  } else {
    load_b(0, 0);
    store_halo(Set0{});
    store_b(0);
    load_b(0, 1);              // SPI >= 3
  }
  if constexpr (DEEPH) load_next_halo(Set1{});   // item 1 -> set 1
  int c = 0, pi = 0;         // chunk / patch of the item being computed
  int par = 0;               // weight buffer of the step being computed
  [[maybe_unused]] int st = 0;
#if defined(MGU_DIAG) && MGU_DIAG == 23
  const bool diag_on = blockIdx.x == 40 && blockIdx.y == 0 && d.H == MGU_DIAG_H && d.Cp == MGU_DIAG_CP && d.N == MGU_DIAG_N && sizeof(T) == 2;
#endif
  // The SPI steps of an item are unrolled (the tap is a compile-time constant) and EVERY step issues the same loads whatever the
  // position in the walk (past the end: clamped re-reads that nobody uses): with a runtime tap and conditional loads hipcc merged the
  // pending-load states of the paths into `s_waitcnt vmcnt(0)` at every step, and -- vmcnt retiring in order -- each weight tile's wait
  // was a wait for the HBM halo loads issued one step earlier (a timing build without them: -13 % on the bf16 forward).  Order
  // inside a step: MFMAs, then the next step's weight tile goes to LDS (requested one step ago), then the tile after it is
  // requested, then (step 0) the halo of the next item (DEEPH: of the item after it, into the set of this item's parity): the halo
  // loads are behind the weight loads in the queue, so the first wait that covers them is the one at the bottom of step 2.
  auto do_item = [&](auto par_c) {
    using SetLoad = std::integral_constant<int, DEEPH ? decltype(par_c)::value : 0>;        // receives item + 2 (item + 1)
    using SetStore = std::integral_constant<int, DEEPH ? (decltype(par_c)::value ^ 1) : 0>;  // holds item + 1
    const int cnext = c + 1 == nchunks ? 0 : c + 1;   // chunk of the next item
    x3_static_for<0, SPI>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      HALO_T(0);
      if constexpr (GLDS) {
        // this step's weight tile has landed: everything of this wave older than the operations issued BEHIND its DMA is complete.
        // NBUF == 2: the DMA was issued at the top of the previous step; behind it: the halo loads of step 0 (tap 1), the previous
        // patch's stores (tap 0).  NBUF == 3: issued two steps ago; behind it additionally the DMA of the step in between (NGL) and the
        // halo loads when one of the two steps was step 0.  Counts are lower bounds of what is really in flight (vmcnt counts to 63).
        constexpr int HRL = HR < 63 ? HR : 63;
        if constexpr (NBUF == 2) {
          if constexpr (tap == 1) wait_vmcnt<HRL>();
          else if constexpr (tap == 0) {
            if (w0) wait_vmcnt<NST>();
            else wait_vmcnt<0>();
          } else wait_vmcnt<0>();
        } else {
          constexpr int both = NGL + HRL < 63 ? NGL + HRL : 63;
          if constexpr (tap == 1 || tap == 2) wait_vmcnt<both>();
          else wait_vmcnt<NGL>();   // (behind a patch's epilogue this also waits for its stores: once per patch)
        }
      }
      wg_barrier();  // Bs[par] (and a fresh halo when tap == 0) visible; the buffer of the previous step no longer read
      HALO_T(1);
#if defined(MGU_DIAG) && MGU_DIAG == 50
      // one-shot checking build (never shipped): the weight tile of THIS step, as it stands in LDS behind the hand-counted wait and
      // the barrier, against its global source; a mismatch sets bit 0 of mgu_diag_glds_bad, every compared 16-byte piece counts in
      // mgu_diag_glds_n (tests read both through mgu_diag_glds_read)
      if constexpr (GLDS) {
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
          const int k0 = (tap * TPS + tt) * d.Cp + c * CK;
          for (int e = tid; e < BN * 8; e += 256) {
            const int row = e >> 3, piece = e & 7;
            const int pos = piece ^ ((row >> 1) & 7);      // LDS position of logical piece `piece` of the row
            const u32x4 got = *reinterpret_cast<const u32x4*>(Bs + ((size_t)(par * TPS + tt) * BN + row) * BROW + pos * VEC);
            const u32x4 ref = *reinterpret_cast<const u32x4*>(w_t + (size_t)(bn0 + row) * d.Kp + k0 + piece * VEC);
            if (got[0] != ref[0] || got[1] != ref[1] || got[2] != ref[2] || got[3] != ref[3]) atomicOr(&mgu_diag_glds_bad, 1u);
          }
          if (tid == 0) atomicAdd(&mgu_diag_glds_n, (unsigned long long)(BN * 8));
        }
      }
#endif
      if constexpr (GLDS) {   // the tile of step s + NBUF - 1 into the buffer the previous step read
        constexpr int ahead = NBUF - 1;
        const int tgt = NBUF == 2 ? (par ^ 1) : (par == 0 ? 2 : par - 1);
        if constexpr (tap + ahead < SPI) glds_b(c, tap + ahead, tgt);
        else glds_b(cnext, tap + ahead - SPI, tgt);
        if constexpr (tap == 0) load_next_halo(SetLoad{});   // behind the DMA in the queue
      }
    // operand fetches are software-pipelined one (tap, kk) group ahead of the MFMAs that consume them, so the
    // ~100-cycle ds_read latency hides under the previous group's MFMAs instead of being exposed 4x per tap
    {
      constexpr int NKK = NP / 2;               // 32-byte k groups per chunk
      constexpr int NG = TPS * NKK;
      // the operands are requested two k groups ahead (a ring of three fragment sets) where the registers allow: alone on its SIMD a
      // wave took 930 cycles for the 512 of a step's 16 MFMAs with a one-group lead (timeline of a one-workgroup-per-CU run) -- an
      // LDS round trip per group.  (All four groups up front: 255 registers + spills.)
      constexpr int DEPTH = (GLDS && NG == 4 && HR <= 6) ? 4 : (NG >= 3 && HR * NSETH <= 12) ? 3 : 2;   // (the 16 x 16 pixel tiles with 64-channel chunks hold 11 halo registers x 4: no room)
      f32x4 a[DEPTH][WMT], b[DEPTH][WNT];
      auto fetch = [&](int gidx, int slot) {
        const int tt = gidx / NKK, kk = gidx % NKK;
        const int tp = tap * TPS + tt;            // 3x3 tap index
        const int r = tp / 3, s = tp - 3 * r;
        const T* Ap = Hs + (r * HWID + s) * LDS_LD + kk * 2 * VEC;
        const T* Bp = GLDS ? Bs + (par * TPS + tt) * BN * BROW + boff + (((2 * kk + lh) ^ bswz) * VEC)
                           : Bs + (par * TPS + tt) * BN * LDS_LD + boff + kk * 2 * VEC;
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) a[slot][mi] = *reinterpret_cast<const f32x4*>(Ap + aoff[mi]);
#pragma unroll
        for (int ni = 0; ni < WNT; ++ni) b[slot][ni] = *reinterpret_cast<const f32x4*>(Bp + ni * 32 * BROW);
      };
#pragma unroll
      for (int gidx = 0; gidx < DEPTH - 1; ++gidx) fetch(gidx, gidx);
      __builtin_amdgcn_sched_barrier(0);   // (hipcc sinks the reads back in front of their MFMAs otherwise: fewer live registers)
#pragma unroll
      for (int gidx = 0; gidx < NG; ++gidx) {
        if (gidx + DEPTH - 1 < NG) {
          fetch(gidx + DEPTH - 1, (gidx + DEPTH - 1) % DEPTH);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
          for (int ni = 0; ni < WNT; ++ni) acc[mi][ni] = mma_chunk<T>(a[gidx % DEPTH][mi], b[gidx % DEPTH][ni], acc[mi][ni]);
      }
    }
      HALO_T(2);   // MFMAs of the step issued
      if constexpr (!GLDS) {
        store_b(par ^ 1);                         // the next step's weight tile (its buffer is free since the barrier above)
        if constexpr (tap + 2 < SPI) load_b(c, tap + 2);   // the tile of the step after it: (chunk, step) two steps ahead
        else load_b(cnext, tap + 2 - SPI);
        if constexpr (tap == 0) load_next_halo(SetLoad{});
      }
      if constexpr (NBUF == 2) par ^= 1;
      else par = par == 2 ? 0 : par + 1;
      HALO_T(3);
      ++st;
    });
    {
      const bool patch_done = (c + 1 == nchunks);
      const bool more = patch_done ? (pi + 1 < npatch) : true;
      if (more) {
        wg_barrier();  // every wave is done with the old halo
        store_halo(SetStore{});     // visible after the barrier at the top of the next step
      }
      w0 = 0;
      if (patch_done) {
        // ---- epilogue of patch pi (its stores overlap the next patch's halo, already in LDS/flight) ----
        const PatchPos pp = setup_patch(p_begin + pi);
        const int img = pp.img, y0 = pp.y0, x0 = pp.x0;
        // uniform 64-bit image base + 32-bit element index; per element only one add (row/col constants fold
        // into scalar multiples of W*ldout and ldout), bounds tests only on patches that cross the image edge
        T* const img_out = reinterpret_cast<T*>(d.out) + (size_t)img * d.H * d.W * d.ldout + d.coff;
        // block-uniform fast path: a patch inside the image with whole n tiles stores with NO per-element condition
        // (hipcc puts an s_waitcnt vmcnt(0) in front of every conditional store: serialised store round trips)
        const bool interior = (y0 + TH <= d.H) && (x0 + TW <= d.W) && (bn0 + BN <= d.N);
        const unsigned sA = (unsigned)(d.W * d.ldout), sB = (unsigned)d.ldout;
        // interior patches store through a buffer descriptor: the lane part of the address (pixel column half + channel) is ONE
        // 32-bit VGPR offset per n tile and the accumulator register's pixel rides in the SCALAR offset (computed on the SALU), so a
        // stored value costs no address VALU at all (the flat form took an add and a 64-bit shift-add per value: with scale, ReLU and
        // the conversion 6-7 VALU per value, 17 % of a 64-channel layer's patch time on the two waves of a SIMD)
        const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(img_out, 0, 0x7ffffff0, 0x00020000);
        const int wm_s = __builtin_amdgcn_readfirstlane(wm);   // wave-uniform by construction: tell the compiler (the scalar offset below)
        // y = max(relu_lo, y): one instruction whether the layer has a ReLU or not.  Without a ReLU the bound is a quiet NaN: v_max_f32
        // then returns y for EVERY y, a NaN included (with -inf as the bound a NaN accumulator was stored as -inf and no longer showed
        // in the output)
        const float relu_lo = d.relu ? 0.f : __builtin_nanf("");
        auto store_patch = [&](auto guarded_t) {
          constexpr bool GUARDED = decltype(guarded_t)::value;
#pragma unroll
          for (int ni = 0; ni < WNT; ++ni) {
            const int n = bn0 + (wn * WNT + ni) * 32 + lr;
            const bool nvalid = n < d.N;
            const float sc = (nvalid && d.scale) ? d.scale[n] : 1.f;
            const float sh = (nvalid && d.shift) ? d.shift[n] : 0.f;
            const unsigned lane_idx = (unsigned)((y0 * d.W + x0 + 4 * lh) * d.ldout + n);
#pragma unroll
            for (int mi = 0; mi < WMT; ++mi) {
              float vv[16];
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) {
                // pixel of this accumulator register inside the patch: pp = 32*(wm*WMT+mi) + 8*(rr>>2) + 4*lh + (rr&3)
                const int pyc = 2 * (wm * WMT + mi) + (rr >> 3);            // patch row    (lane independent)
                const int pxc = 8 * ((rr >> 2) & 1) + (rr & 3);             // patch column (without the 4*lh part)
                float v = acc[mi][ni][rr] * sc + sh;
                v = max_1op(relu_lo, v);
                vv[rr] = v;
                const unsigned idx = lane_idx + (unsigned)pyc * sA + (unsigned)pxc * sB;
                if (!GUARDED) {
                  const int soff = (int)(((unsigned)(2 * (wm_s * WMT + mi) + (rr >> 3)) * sA + (unsigned)pxc * sB) * (unsigned)sizeof(T));
                  if constexpr (sizeof(T) == 2)
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (T)v), out_rsrc, (int)(lane_idx * 2u), soff, 0);
                  else
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, (int)(lane_idx * 4u), soff, 0);
                } else if (nvalid && y0 + pyc < d.H && x0 + pxc + 4 * lh < d.W) {
                  img_out[idx] = (T)v;
                }
                acc[mi][ni][rr] = 0.f;
              }
              if (d.pool) {
                // fused MaxPool2d(2): registers rr, rr^1 (x neighbour) and rr^8 (y neighbour) of a lane are one pooling
                // window (patch origins are even), so the pooled tensor costs four max operations per window and no
                // second pass over the feature map
                T* const pool_img = reinterpret_cast<T*>(d.pool) + (size_t)img * (d.H >> 1) * (d.W >> 1) * d.ldpool;
#pragma unroll
                for (int a = 0; a < 8; a += 2) {   // rr with bit 0 and bit 3 clear: 0, 2, 4, 6
                  const float m = max_1op(max_1op(vv[a], vv[a + 1]), max_1op(vv[a + 8], vv[a + 9]));
                  const int pyc = 2 * (wm * WMT + mi), pxc = 8 * ((a >> 2) & 1) + (a & 3) + 4 * lh;
                  const int py = (y0 + pyc) >> 1, px = (x0 + pxc) >> 1;
                  if (!GUARDED || (nvalid && y0 + pyc + 1 < d.H && x0 + pxc + 1 < d.W))
                    pool_img[((size_t)py * (d.W >> 1) + px) * d.ldpool + n] = (T)m;
                }
              }
            }
          }
        };
#if defined(MGU_DIAG) && MGU_DIAG == 7   // diagnostic build: no output stores (accumulators kept live)
        {
          float sacc = 0.f;
#pragma unroll
          for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
            for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) sacc += acc[mi][ni][rr], acc[mi][ni][rr] = 0.f;
          if (sacc == 123.456f) img_out[tid] = (T)sacc;
        }
#else
        if (interior) store_patch(std::false_type{});
        else store_patch(std::true_type{});
        w0 = interior ? 1 : 0;
#endif
      }
      // (plain arithmetic: `if (done) { c = 0; ++pi; } else ++c;` became an increment through a selected ADDRESS, with c and pi in scratch)
      pi += patch_done ? 1 : 0;
      c = patch_done ? 0 : c + 1;
    }
  };
  for (int item = 0; item < nitems; item += NSETH) {
    do_item(Set0{});
    if constexpr (DEEPH) {
      if (item + 1 < nitems) do_item(Set1{});
    }
  }
}

template <typename T, int NP, int TH, int WAVES_M, int WAVES_N, int WMT, int WNT, int TPS>
static hipError_t launch_halo(const IgemmDesc& d, hipStream_t s) {
  constexpr int BN = WAVES_N * WNT * 32;
  constexpr int HP = (TH + 2) * 18;
  const int tiles_x = (d.W + 15) / 16, tiles_y = (d.H + TH - 1) / TH;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B, ntn = (d.N + BN - 1) / BN;
  // patches per workgroup: keep >= ~4 workgroups per CU in the grid (2 are resident), at most 16 patches each
  int ppb = (int)(((long)total * ntn) / (256 * 4));
  if (ppb < 1) ppb = 1;
  if (ppb > tun(d).halo_max_ppb) ppb = std::max(1, tun(d).halo_max_ppb);
  dim3 grid((total + ppb - 1) / ppb, ntn);
  size_t lds = (size_t)(HP + 2 * TPS * BN) * (NP * 16 + 16);
#if !defined(MGU_HALO_NO_GLDS)
  if (NP == 8 && BN % 32 == 0 && (size_t)HP * (NP * 16 + 16) + (size_t)3 * TPS * BN * NP * 16 <= 80 * 1024)
    lds = (size_t)HP * (NP * 16 + 16) + (size_t)3 * TPS * BN * NP * 16;   // three unpadded weight buffers (the kernel's NBUF)
#endif
#if defined(MGU_DIAG) && MGU_DIAG == 23
  if (getenv("MGU_DIAG_OCC1")) lds = std::max<size_t>(lds, 100 * 1024);   // timing experiment: one workgroup per CU
#endif
  static bool attr_done[64] = {};
  hipError_t ae = ensure_dyn_lds(reinterpret_cast<const void*>(&conv3x3_halo_kernel<T, NP, TH, WAVES_M, WAVES_N, WMT, WNT, TPS>), lds, attr_done);
  if (ae != hipSuccess) return ae;
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, NP, TH, WAVES_M, WAVES_N, WMT, WNT, TPS>), grid, dim3(256), lds, s, d, tiles_x,
                     tiles_y, total, ppb, tun(d).wino_yfast ? 1 : 0);
  return hipGetLastError();
}

template <typename T>
static int halo_np(const IgemmDesc& d) {   // 16-byte pieces per pixel chunk the halo kernel can use, 0 = not applicable
  constexpr int VEC = Elem<T>::VEC;
  // one image in BYTES below the buffer descriptors' reach: the input offsets must stay under the out-of-image marker (0x7fff0000), the
  // interior stores under the output descriptor's num_records (0x7ffffff0) -- a store past it would be dropped silently
  if (!(d.KS == 3 && d.out_mode == 0 && d.K == 9 * d.Cp && d.ldin == d.Cp && (long)d.H * d.W * d.ldin * (long)sizeof(T) < 0x7fff0000l &&
        (long)d.H * d.W * d.ldout * (long)sizeof(T) < 0x7ffffff0l && tun(d).use_halo))
    return 0;
  if (d.Cp % (8 * VEC) == 0) return 8;
  if (sizeof(T) == 2 && d.Cp % (4 * VEC) == 0) return 4;   // bf16 layers with 32 input channels
  return 0;
}

template <typename T, int KS, int OUTMODE, int WAVES_M, int WAVES_N, int WMT, int WNT>
static hipError_t launch_cfg(const IgemmDesc& d, hipStream_t s) {
  constexpr int BM = WAVES_M * WMT * 32;
  constexpr int BN = WAVES_N * WNT * 32;
  dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN);
  const size_t lds = (size_t)(BM + BN) * 144;
  hipLaunchKernelGGL((igemm_kernel<T, KS, OUTMODE, WAVES_M, WAVES_N, WMT, WNT>), grid, dim3(256), lds, s, d);
  return hipGetLastError();
}

template <typename T, int KS, int OUTMODE>
static hipError_t launch_tiles(const IgemmDesc& d, hipStream_t s) {
  if (d.N > 64) return launch_cfg<T, KS, OUTMODE, 2, 2, 2, 2>(d, s);   // 128 x 128 tile, wave 64x64
  if (d.N > 32) return launch_cfg<T, KS, OUTMODE, 4, 1, 2, 2>(d, s);   // 256 x 64 tile,  wave 64x64
  return launch_cfg<T, KS, OUTMODE, 4, 1, 2, 1>(d, s);                 // 256 x 32 tile,  wave 64x32
}

template <typename T, int NP>
static hipError_t launch_halo_tiles(const IgemmDesc& d, hipStream_t s) {
  if (d.N > 64) return launch_halo<T, NP, 8, 2, 2, 2, 2, 1>(d, s);     // 8x16 px  x 128 ch, wave 64x64
  if (d.N > 32) return launch_halo<T, NP, 16, 4, 1, 2, 2, 1>(d, s);    // 16x16 px x 64 ch,  wave 64x64
  if (tun(d).halo_tps3) return launch_halo<T, NP, 16, 4, 1, 2, 1, 3>(d, s);  // 16x16 px x 32 ch, wave 64x32, 3 taps per barrier
  return launch_halo<T, NP, 16, 4, 1, 2, 1, 1>(d, s);
}

// does a descriptor run on the halo kernel (whose epilogue can also write the 2x2 max-pooled tensor, IgemmDesc::pool)?
bool halo_pool_fusable(const IgemmDesc& d, int dtype) {
  if (d.out_mode != 0 || d.split_n) return false;
  return dtype == 0 ? (!wino_applicable(d) && halo_np<float>(d) == 8) : halo_np<__bf16>(d) != 0;
}

const char* igemm_kernel_name(const IgemmDesc& d, int dtype) {
  if (dtype == 1) {
    if (d.out_mode == 1) return convt_bf16f_applicable(d) ? "convt2x2_bf16_kernel" : "igemm_kernel<bf16> (ConvTranspose)";
    return halo_np<__bf16>(d) ? "conv3x3_halo_kernel<bf16>" : "igemm_kernel<bf16>";
  }
  if (d.out_mode == 1) return convt_x3_applicable(d) ? "convt2x2_x3_kernel" : "igemm_kernel<f32> (ConvTranspose)";
  if (wino_applicable(d)) {
    const bool wide = d.N > 32 && tun(d).wino_mode != 1;
    if (tun(d).wino_prec && tun(d).wino_cp && (wide || tun(d).wino_cp_narrow) && (long)d.H * d.W * d.ldin * 4 < (1l << 31))
      return wide ? (wino_asm_applicable(d) ? "mgu_wino_cp2_gfx950 (asm form of wino3x3_cp_kernel<2>)" : "wino3x3_cp_kernel<2>")
                  : (wino_asm_applicable(d) ? "mgu_wino_cp1r_gfx950 (asm form of wino3x3_cp_kernel<1>)" : "wino3x3_cp_kernel<1>");
    if (tun(d).wino_prec) return wide ? "wino3x3_f32_kernel<0,1>" : "wino3x3_f32_kernel<1,1>";
    return wide ? "wino3x3_f32_kernel<0,0>" : "wino3x3_f32_kernel<1,0>";
  }
  if (halo_np<float>(d) == 8) return "conv3x3_halo_kernel<f32>";
  if (d.KS == 2) return convt_x3_dgrad_applicable(d) ? "convt2x2_x3_kernel<dgrad>" : "igemm_kernel<f32> (ConvTranspose dgrad)";
  return "igemm_kernel<f32>";
}

hipError_t launch_igemm_f32(const IgemmDesc& d, hipStream_t s) {
  if (d.M <= 0 || d.N <= 0) return hipSuccess;
  if ((d.Cp & 3) || (d.ldin & 3) || (d.Kp % 32) || d.K > d.Kp) return hipErrorInvalidValue;
  if (d.out_mode == 1) {
    if (d.KS != 1) return hipErrorInvalidValue;
    if (convt_x3_applicable(d)) return launch_convt_x3(d, s);
    return launch_tiles<float, 1, 1>(d, s);
  }
  if (d.KS == 2 && convt_x3_dgrad_applicable(d)) return launch_convt_x3_dgrad(d, s);
  if (wino_applicable(d)) return launch_wino_f32(d, s);
  if (halo_np<float>(d) == 8) return launch_halo_tiles<float, 8>(d, s);
  if (d.KS == 3) return launch_tiles<float, 3, 0>(d, s);
  if (d.KS == 1) return launch_tiles<float, 1, 0>(d, s);
  if (d.KS == 2) return launch_tiles<float, 2, 0>(d, s);
  return hipErrorInvalidValue;
}

// bf16 storage, fp32 accumulate (inference): `in`, `w`, `out` of the descriptor point to bf16 data, ld/Cp/K/Kp are in
// elements (Cp % 8 == 0, Kp % 64 == 0); scale/shift stay fp32.
hipError_t launch_igemm_bf16(const IgemmDesc& d, hipStream_t s) {
  if (d.M <= 0 || d.N <= 0) return hipSuccess;
  if ((d.Cp & 7) || (d.ldin & 7) || (d.Kp % 64) || d.K > d.Kp || d.split_n) return hipErrorInvalidValue;
  if (d.out_mode == 1) {
    if (d.KS != 1) return hipErrorInvalidValue;
    if (convt_bf16f_applicable(d)) return launch_convt_bf16f(d, s);
    return launch_tiles<__bf16, 1, 1>(d, s);
  }
  const int np = halo_np<__bf16>(d);
  if (np == 8) return launch_halo_tiles<__bf16, 8>(d, s);
  if (np == 4) return launch_halo_tiles<__bf16, 4>(d, s);
  if (d.KS == 3) return launch_tiles<__bf16, 3, 0>(d, s);
  if (d.KS == 1) return launch_tiles<__bf16, 1, 0>(d, s);
  return hipErrorInvalidValue;
}

}  // namespace mgu

#if defined(MGU_DIAG) && MGU_DIAG == 23
extern "C" int mgu_diag_read(unsigned long long* out, int n) {
  if (n > 4 * 256 * 4) n = 4 * 256 * 4;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgu::mgu_halo_ts), (size_t)n * sizeof(unsigned long long));
}
#endif

#if defined(MGU_DIAG) && MGU_DIAG == 50
// checking build: (mismatch flag, 16-byte pieces compared) of the LDS-DMA weight tiles since the last call; clears both
extern "C" int mgu_diag_glds_read(unsigned* bad, unsigned long long* n) {
  unsigned z = 0;
  unsigned long long zn = 0;
  if (hipMemcpyFromSymbol(bad, HIP_SYMBOL(mgu::mgu_diag_glds_bad), sizeof(unsigned)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(mgu::mgu_diag_glds_n), sizeof(unsigned long long)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mgu::mgu_diag_glds_bad), &z, sizeof z) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mgu::mgu_diag_glds_n), &zn, sizeof zn) != hipSuccess) return -1;
  return 0;
}
#endif
