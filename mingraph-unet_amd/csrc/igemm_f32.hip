// Implicit-GEMM convolution / transposed-convolution / linear kernel for gfx950 (CDNA4), fp32.
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

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static bool g_use_halo = true;   // MGU_NO_HALO=1 forces the generic gather kernel (A/B comparisons)
void set_use_halo(bool on) { g_use_halo = on; }

constexpr int CK = 32;           // K elements per pipeline step
constexpr int LDS_LD = CK + 4;   // LDS row pitch in floats (144 B, keeps 16-B alignment)

template <int KS, int OUTMODE, int WAVES_M, int WAVES_N, int WMT, int WNT>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IgemmDesc d) {
  static_assert(WAVES_M * WAVES_N == 4, "4 wavefronts per workgroup");
  constexpr int BM = WAVES_M * WMT * 32;
  constexpr int BN = WAVES_N * WNT * 32;
  constexpr int AR = BM / 32;  // A rows staged per thread
  constexpr int BR = BN / 32;  // B rows staged per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + BM * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int bm0 = blockIdx.x * BM;
  const int bn0 = blockIdx.y * BN;
  const int kq = tid & 7;    // which float4 of the 32-wide K slice this thread stages
  const int r0 = tid >> 3;   // first tile row this thread stages (then +32, +64, ...)

  // ---- per-row gather state: pixel base pointer + 9-bit tap validity mask ---------------------
  const float* abase[AR];
  unsigned amask[AR];
  const int HW = d.H * d.W;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = bm0 + r0 + 32 * i;
    abase[i] = d.in;
    amask[i] = 0u;
    if (m < d.M) {
      abase[i] = d.in + (size_t)m * d.ldin;
      if (KS == 2) {
        // stride-2 2x2 gather (ConvTranspose2d dgrad): row m = (img, y, x) over H x W reads the four
        // pixels (2y+dy, 2x+dx) of the Hout x Wout source grid; every tap is in range (Hout >= 2H).
        const int img = m / HW;
        const int rem = m - img * HW;
        const int oy = rem / d.W;
        const int ox = rem - oy * d.W;
        abase[i] = d.in + (((size_t)img * d.Hout + 2 * oy) * d.Wout + 2 * ox) * d.ldin;
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
  const float* wrow[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) wrow[i] = d.w + (size_t)(bn0 + r0 + 32 * i) * d.Kp + kq * 4;

  f32x4 areg[AR];
  f32x4 breg[BR];
  auto load_tiles = [&](int k0) {
    const int k = k0 + kq * 4;
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
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kvalid && ((amask[i] >> tap) & 1u)) v = *reinterpret_cast<const f32x4*>(abase[i] + delta);
      areg[i] = v;
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
  const float* Ap = As + (wm * WMT * 32 + lr) * LDS_LD + lh * 4;
  const float* Bp = Bs + (wn * WNT * 32 + lr) * LDS_LD + lh * 4;
  float* Asw = As + r0 * LDS_LD + kq * 4;
  float* Bsw = Bs + r0 * LDS_LD + kq * 4;

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
    for (int kk = 0; kk < CK / 8; ++kk) {
      if (ks * CK + kk * 8 >= d.K) break;  // zero K tail (block-uniform): nothing to accumulate
      f32x4 a[WMT], b[WNT];
#pragma unroll
      for (int mi = 0; mi < WMT; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(Ap + mi * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int ni = 0; ni < WNT; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Bp + ni * 32 * LDS_LD + kk * 8);
      // lane (lr, lh) holds k = kk*8 + lh*4 + t for t = 0..3: MFMA t contracts k(lh=0) and k(lh=1);
      // A and B use the same (lh, t) -> k map, so the permuted k order is consistent.
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
          for (int ni = 0; ni < WNT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int ni = 0; ni < WNT; ++ni) {
    const int n = bn0 + (wn * WNT + ni) * 32 + lr;
    const bool nvalid = n < d.N;
    const float sc = (nvalid && d.scale) ? d.scale[n] : 1.f;
    const float sh = (nvalid && d.shift) ? d.shift[n] : 0.f;
    int q = 0, co = n;
    if (OUTMODE == 1) {
      q = n / d.ct_cout;
      co = n - q * d.ct_cout;
    }
#pragma unroll
    for (int mi = 0; mi < WMT; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = bm0 + (wm * WMT + mi) * 32 + row;
        if (m < d.M && nvalid) {
          float v = acc[mi][ni][r] * sc + sh;
          if (d.relu) v = fmaxf(v, 0.f);
          if (OUTMODE == 0) {
            if (d.split_n > 0 && n >= d.split_n) d.out2[(size_t)m * d.ld2 + (n - d.split_n)] = v;
            else d.out[(size_t)m * d.ldout + d.coff + n] = v;
          } else {
            const int img = m / HW;
            const int rem = m - img * HW;
            const int y = rem / d.W;
            const int x = rem - y * d.W;
            const size_t pix = ((size_t)img * d.Hout + (2 * y + (q >> 1))) * d.Wout + (2 * x + (q & 1));
            d.out[pix * d.ldout + d.coff + co] = v;
          }
        }
      }
    }
  }
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
template <int TH, int WAVES_M, int WAVES_N, int WMT, int WNT>
__global__ __launch_bounds__(256) void conv3x3_halo_f32_kernel(const IgemmDesc d, const int tiles_x, const int tiles_y) {
  constexpr int TW = 16, HWID = TW + 2, HP = (TH + 2) * HWID;
  constexpr int BM = TH * TW;
  constexpr int BN = WAVES_N * WNT * 32;
  static_assert(BM == WAVES_M * WMT * 32 && WAVES_M * WAVES_N == 4, "tile/wave mismatch");
  constexpr int HR = (HP + 31) / 32;  // halo pixels staged per thread
  constexpr int BR = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                  // [HP][36]
  float* Bs = smem + HP * LDS_LD;    // [2][BN][36]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int kq = tid & 7;
  const int r0 = tid >> 3;
  const int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  const int ty = (bid / tiles_x) % tiles_y;
  const int img = bid / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  const int bn0 = blockIdx.y * BN;
  const float* img_base = d.in + (size_t)img * d.H * d.W * d.ldin;

  // halo staging map: thread -> (halo pixel r0 + 32 i, float4 kq); offset < 0: outside the image (zero)
  int hoff[HR];
#pragma unroll
  for (int i = 0; i < HR; ++i) {
    const int hp = r0 + 32 * i;
    const int hy = hp / HWID, hx = hp - hy * HWID;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    hoff[i] = (hp < HP && y >= 0 && y < d.H && x >= 0 && x < d.W) ? (y * d.W + x) * d.ldin + kq * 4 : -1;
  }
  const float* wrow[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) wrow[i] = d.w + (size_t)(bn0 + r0 + 32 * i) * d.Kp + kq * 4;

  f32x4 hreg[HR];
  f32x4 breg[BR];
  auto load_halo = [&](int c) {
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (hoff[i] >= 0) v = *reinterpret_cast<const f32x4*>(img_base + hoff[i] + c * CK);
      hreg[i] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < HR; ++i)
      if (r0 + 32 * i < HP) *reinterpret_cast<f32x4*>(Hs + (r0 + 32 * i) * LDS_LD + kq * 4) = hreg[i];
  };
  auto load_b = [&](int c, int tap) {
    const int k0 = tap * d.Cp + c * CK;
#pragma unroll
    for (int i = 0; i < BR; ++i) breg[i] = *reinterpret_cast<const f32x4*>(wrow[i] + k0);
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BR; ++i)
      *reinterpret_cast<f32x4*>(Bs + buf * BN * LDS_LD + (r0 + 32 * i) * LDS_LD + kq * 4) = breg[i];
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
    const int p = (wm * WMT + mi) * 32 + lr;
    aoff[mi] = ((p >> 4) * HWID + (p & 15)) * LDS_LD + lh * 4;
  }
  const int boff = (wn * WNT * 32 + lr) * LDS_LD + lh * 4;

  const int nchunks = d.Cp / CK;
  const int nsteps = nchunks * 9;
  load_halo(0);
  load_b(0, 0);
  store_halo();
  store_b(0);
  if (nsteps > 1) load_b(0, 1);
  int c = 0, tap = 0;
  for (int st = 0; st < nsteps; ++st) {
    __syncthreads();  // Bs[st&1] (and a fresh halo when tap == 0) visible; Bs[(st+1)&1] no longer read
    if (st + 1 < nsteps) store_b((st + 1) & 1);
    if (st + 2 < nsteps) {
      int t2 = tap + 2, c2 = c;
      if (t2 >= 9) { t2 -= 9; ++c2; }
      load_b(c2, t2);
    }
    if (tap == 0 && c + 1 < nchunks) load_halo(c + 1);
    const int r = tap / 3, s = tap - 3 * r;
    const float* Ap = Hs + (r * HWID + s) * LDS_LD;
    const float* Bp = Bs + (st & 1) * BN * LDS_LD + boff;
#pragma unroll
    for (int kk = 0; kk < CK / 8; ++kk) {
      f32x4 a[WMT], b[WNT];
#pragma unroll
      for (int mi = 0; mi < WMT; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(Ap + aoff[mi] + kk * 8);
#pragma unroll
      for (int ni = 0; ni < WNT; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Bp + ni * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi)
#pragma unroll
          for (int ni = 0; ni < WNT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t], b[ni][t], acc[mi][ni], 0, 0, 0);
    }
    if (++tap == 9) {
      tap = 0;
      ++c;
      if (c < nchunks) {
        __syncthreads();  // every wave is done with the old halo
        store_halo();
      }
    }
  }

#pragma unroll
  for (int ni = 0; ni < WNT; ++ni) {
    const int n = bn0 + (wn * WNT + ni) * 32 + lr;
    const bool nvalid = n < d.N;
    const float sc = (nvalid && d.scale) ? d.scale[n] : 1.f;
    const float sh = (nvalid && d.shift) ? d.shift[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < WMT; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int p = (wm * WMT + mi) * 32 + row;
        const int y = y0 + (p >> 4), x = x0 + (p & 15);
        if (nvalid && y < d.H && x < d.W) {
          float v = acc[mi][ni][r] * sc + sh;
          if (d.relu) v = fmaxf(v, 0.f);
          d.out[(((size_t)img * d.H + y) * d.W + x) * d.ldout + d.coff + n] = v;
        }
      }
    }
  }
}

template <int TH, int WAVES_M, int WAVES_N, int WMT, int WNT>
static hipError_t launch_halo(const IgemmDesc& d, hipStream_t s) {
  constexpr int BN = WAVES_N * WNT * 32;
  constexpr int HP = (TH + 2) * 18;
  const int tiles_x = (d.W + 15) / 16, tiles_y = (d.H + TH - 1) / TH;
  const int B = d.M / (d.H * d.W);
  dim3 grid(tiles_x * tiles_y * B, (d.N + BN - 1) / BN);
  const size_t lds = (size_t)(HP + 2 * BN) * LDS_LD * sizeof(float);
  hipLaunchKernelGGL((conv3x3_halo_f32_kernel<TH, WAVES_M, WAVES_N, WMT, WNT>), grid, dim3(256), lds, s, d, tiles_x, tiles_y);
  return hipGetLastError();
}

static bool halo_applicable(const IgemmDesc& d) {
  return d.KS == 3 && d.out_mode == 0 && (d.Cp % CK) == 0 && d.K == 9 * d.Cp && d.ldin == d.Cp &&
         (long)d.H * d.W * d.ldin < (1l << 31) && g_use_halo;
}

template <int KS, int OUTMODE, int WAVES_M, int WAVES_N, int WMT, int WNT>
static hipError_t launch_cfg(const IgemmDesc& d, hipStream_t s) {
  constexpr int BM = WAVES_M * WMT * 32;
  constexpr int BN = WAVES_N * WNT * 32;
  dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN);
  const size_t lds = (size_t)(BM + BN) * LDS_LD * sizeof(float);
  hipLaunchKernelGGL((igemm_f32_kernel<KS, OUTMODE, WAVES_M, WAVES_N, WMT, WNT>), grid, dim3(256), lds, s, d);
  return hipGetLastError();
}

template <int KS, int OUTMODE>
static hipError_t launch_tiles(const IgemmDesc& d, hipStream_t s) {
  if (d.N > 64) return launch_cfg<KS, OUTMODE, 2, 2, 2, 2>(d, s);   // 128 x 128 tile, wave 64x64
  if (d.N > 32) return launch_cfg<KS, OUTMODE, 4, 1, 2, 2>(d, s);   // 256 x 64 tile,  wave 64x64
  return launch_cfg<KS, OUTMODE, 4, 1, 2, 1>(d, s);                 // 256 x 32 tile,  wave 64x32
}

hipError_t launch_igemm_f32(const IgemmDesc& d, hipStream_t s) {
  if (d.M <= 0 || d.N <= 0) return hipSuccess;
  if ((d.Cp & 3) || (d.ldin & 3) || (d.Kp % CK) || d.K > d.Kp) return hipErrorInvalidValue;
  if (d.out_mode == 1) {
    if (d.KS != 1) return hipErrorInvalidValue;
    return launch_tiles<1, 1>(d, s);
  }
  if (halo_applicable(d)) {
    if (d.N > 64) return launch_halo<8, 2, 2, 2, 2>(d, s);    // 8x16 px  x 128 ch, wave 64x64
    if (d.N > 32) return launch_halo<16, 4, 1, 2, 2>(d, s);   // 16x16 px x 64 ch,  wave 64x64
    return launch_halo<16, 4, 1, 2, 1>(d, s);                 // 16x16 px x 32 ch,  wave 64x32
  }
  if (d.KS == 3) return launch_tiles<3, 0>(d, s);
  if (d.KS == 1) return launch_tiles<1, 0>(d, s);
  if (d.KS == 2) return launch_tiles<2, 0>(d, s);
  return hipErrorInvalidValue;
}

}  // namespace mgu
