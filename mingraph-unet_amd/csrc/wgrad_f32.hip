// Weight-gradient GEMM for gfx950, fp32 MFMA: Dw[n][k] += sum_m Z[m][n] * A(m,k).
//
// Backward of nn.Conv2d / nn.ConvTranspose2d w.r.t. the weight (the autograd node behind
// loss.backward() at scripts/train_segmentation.py:133).  Z is the gradient flowing into the
// convolution's output (rows = output pixels), A(m,k) is the same on-the-fly im2col gather the forward
// kernel uses (k = tap*Cp + c), so no column buffer is ever materialised.  The reduction runs over the
// PIXEL index m, which is the slow (row) index of both operands in NHWC memory; with the 32x32x2 fp32
// MFMA a lane supplies ONE float per operand (A[i = lane&31][kk = lane>>5]), so operands are fetched
// from m-major LDS tiles with conflict-free ds_read_b32 -- no transpose anywhere.
// The pixel range is split over blockIdx.y; every split writes a PRIVATE partial panel with plain stores (two contiguous
// 128-byte row segments per wave instruction) and the unpack kernel adds the panels in a fixed order: float atomics retire at
// only ~1.3 TB/s chip-wide (MI355X_MICROARCH.md "Global float atomics") and their summation order changed the last bit of a
// weight gradient from run to run -- with train-mode BatchNorm + MaxPool near-ties downstream, that made two identical train
// steps diverge (round-2 review); now the whole backward is bitwise reproducible.
#include "common.h"

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_MK = 32;    // pixels reduced per pipeline step
constexpr int WG_BK = 128;   // k columns per workgroup

template <int KS, int WAVES_N, int WAVES_K, int WNT, int WKT>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradDesc d) {
  static_assert(WAVES_N * WAVES_K == 4, "4 wavefronts");
  constexpr int BN = WAVES_N * WNT * 32;
  static_assert(WAVES_K * WKT * 32 == WG_BK, "k tile is 128");
  constexpr int QZ = BN / 4;           // float4 per Z row
  constexpr int RPZ = 256 / QZ;        // Z rows per pass
  constexpr int PZ = (WG_MK + RPZ - 1) / RPZ;
  constexpr int PA = WG_MK / 8;        // A: 32 float4 per row, 8 rows per pass
  __shared__ __attribute__((aligned(16))) float Zs[WG_MK * BN];
  __shared__ __attribute__((aligned(16))) float As[WG_MK * WG_BK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WAVES_K, wk = wave % WAVES_K;
  const int ktiles = (d.K + WG_BK - 1) / WG_BK;
  const int n0 = (blockIdx.x / ktiles) * BN;
  const int k0 = (blockIdx.x % ktiles) * WG_BK;
  const int m_begin = blockIdx.y * d.rows_per_split;
  const int m_end = min(d.M, m_begin + d.rows_per_split);
  if (m_begin >= m_end) return;

  // ---- A gather: this thread always stages float4 column ja of rows ra + 8 i -----------------------
  const int ja = tid & 31, ra = tid >> 5;
  const int k = k0 + ja * 4;
  const bool kvalid = k < d.K;
  int dy = 0, dx = 0, cch = k;
  if (KS != 1) {
    const int tap = k / d.Cp;
    cch = k - tap * d.Cp;
    dy = tap / KS;
    dx = tap - dy * KS;
    if (KS == 3) { dy -= 1; dx -= 1; }
  }
  int a_img[PA], a_y[PA], a_x[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m_begin + ra + 8 * i;
    const int HW = d.H * d.W;
    a_img[i] = m / HW;
    const int rem = m - a_img[i] * HW;
    a_y[i] = rem / d.W;
    a_x[i] = rem - a_y[i] * d.W;
  }
  // ---- Z: float4 column jz of rows rz + RPZ i -------------------------------------------------------
  const int jz = tid % QZ, rz = tid / QZ;
  const bool zcol_ok = (n0 + jz * 4) < d.N;   // N % 4 == 0 is required by the launcher

  f32x4 areg[PA], zreg[PZ];
  auto load_step = [&](int mb) {
    // unconditional loads from a safe address + select (a branch around a load makes hipcc drain vmcnt per element)
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = mb + ra + 8 * i;
      int sy, sx, SH, SW;
      if (KS == 2) { sy = 2 * a_y[i] + dy; sx = 2 * a_x[i] + dx; SH = d.Hs; SW = d.Ws; }
      else         { sy = a_y[i] + dy;     sx = a_x[i] + dx;     SH = d.H;  SW = d.W; }
      const bool ok = kvalid && m < m_end && sy >= 0 && sy < SH && sx >= 0 && sx < SW;
      const float* src = ok ? d.in + (((size_t)a_img[i] * SH + sy) * SW + sx) * d.ldin + d.inoff + cch : d.in;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      areg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < PZ; ++i) {
      const int r = rz + RPZ * i;
      const int m = mb + r;
      const bool ok = zcol_ok && r < WG_MK && m < m_end;
      const float* src = ok ? d.z + (size_t)m * d.ldz + d.zoff + n0 + jz * 4 : d.z;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      zreg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto advance_rows = [&]() {  // rows move 32 pixels forward: incremental (img, y, x) update
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      a_x[i] += WG_MK;
      while (a_x[i] >= d.W) {
        a_x[i] -= d.W;
        if (++a_y[i] >= d.H) { a_y[i] = 0; ++a_img[i]; }
      }
    }
  };

  f32x16 acc[WNT][WKT];
#pragma unroll
  for (int a = 0; a < WNT; ++a)
#pragma unroll
    for (int b = 0; b < WKT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* Zp = Zs + lh * BN + wn * WNT * 32 + lr;
  const float* Ap = As + lh * WG_BK + wk * WKT * 32 + lr;

  load_step(m_begin);
  for (int mb = m_begin; mb < m_end; mb += WG_MK) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(As + (ra + 8 * i) * WG_BK + ja * 4) = areg[i];
#pragma unroll
    for (int i = 0; i < PZ; ++i)
      if (rz + RPZ * i < WG_MK) *reinterpret_cast<f32x4*>(Zs + (rz + RPZ * i) * BN + jz * 4) = zreg[i];
    __syncthreads();
    if (mb + WG_MK < m_end) {
      advance_rows();
      load_step(mb + WG_MK);
    }
#pragma unroll
    for (int t = 0; t < WG_MK / 2; ++t) {
      float a[WNT], b[WKT];
#pragma unroll
      for (int i = 0; i < WNT; ++i) a[i] = Zp[2 * t * BN + i * 32];
#pragma unroll
      for (int j = 0; j < WKT; ++j) b[j] = Ap[2 * t * WG_BK + j * 32];
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WKT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  float* const dwp = d.dw + (size_t)blockIdx.y * d.N * d.Kp;
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WKT; ++j) {
      const int kk = k0 + (wk * WKT + j) * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * WNT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        // plain stores into the PRIVATE partial panel of this pixel split (blockIdx.y): no float atomics, so the summed
        // gradient (unpack_conv*_grad_kernel adds the panels in a fixed order) is bitwise reproducible from run to run
        if (n < d.N && kk < d.K) dwp[(size_t)n * d.Kp + kk] = acc[i][j][r];
      }
    }
}

template <int KS, int WAVES_N, int WAVES_K, int WNT, int WKT>
static hipError_t launch_wg(WgradDesc& d, hipStream_t s) {
  constexpr int BN = WAVES_N * WNT * 32;
  const int ntiles = (d.N + BN - 1) / BN, ktiles = (d.K + WG_BK - 1) / WG_BK;
  // split the pixel range so the grid has ~4 workgroups per CU, >= 8 steps per split; one partial panel per split
  int splits = (1024 + ntiles * ktiles - 1) / (ntiles * ktiles);
  const size_t cap = d.dw_capacity / ((size_t)d.N * d.Kp);
  if ((size_t)splits > cap) splits = (int)cap;
  int rows = (d.M + splits - 1) / splits;
  if (rows < 8 * WG_MK) rows = 8 * WG_MK;
  rows = (rows + WG_MK - 1) / WG_MK * WG_MK;
  splits = (d.M + rows - 1) / rows;           // every split has >= 1 row and writes every (n < N, k < K) of its panel
  d.rows_per_split = rows;
  d.groups = splits;
  hipLaunchKernelGGL((wgrad_f32_kernel<KS, WAVES_N, WAVES_K, WNT, WKT>), dim3(ntiles * ktiles, splits), dim3(256), 0, s, d);
  return hipGetLastError();
}

template <int KS>
static hipError_t launch_wg_tiles(WgradDesc& d, hipStream_t s) {
  if (d.N > 64) return launch_wg<KS, 2, 2, 2, 2>(d, s);   // 128 (n) x 128 (k), wave 64x64
  if (d.N > 32) return launch_wg<KS, 2, 2, 1, 2>(d, s);   // 64 x 128, wave 32x64
  return launch_wg<KS, 1, 4, 1, 1>(d, s);                 // 32 x 128, wave 32x32
}


// =================================================================================================
// 3x3 weight gradient with an LDS-resident input halo (Cin % 32 == 0, Cout % 32 == 0).
//
// In the generic kernel above every k tile (= tap slice) re-gathers its pixels, and a Cout = 32 layer can
// only do 64 FLOP per gathered float: the full-resolution layers were bound by L2 -> LDS traffic, not by
// the MFMA.  Here a workgroup walks 8 x 16 pixel patches; per patch it stages the dz tile (128 px x NCO*32
// output channels) and the 10 x 18 halo of a 32-channel input chunk ONCE, and all 9 taps x NCO output-channel
// tiles contract over the pixels from LDS (a tap is a constant LDS offset of the B operand).  Each wavefront
// keeps the 9 tap tiles dW[co tile][tap][ci chunk] of one output-channel tile for its share of the pixels in
// 144 accumulator registers across ALL patches of the workgroup; one round of float atomics at the end.
// =================================================================================================
template <int NCO>
__global__ __launch_bounds__(256, 2) void wgrad3x3_halo_f32_kernel(const WgradDesc d, const int tiles_x, const int tiles_y,
                                                               const int total_patches, const int patches_per_block,
                                                               const int nchunks) {
  constexpr int TH = 8, TW = 16, HWID = TW + 2, HP = (TH + 2) * HWID, BM = TH * TW;
  constexpr int HLD = 36;              // halo row pitch (floats)
  constexpr int ZW = NCO * 32;         // dz columns per workgroup
  constexpr int ZLD = ZW + 4;          // dz row pitch
  constexpr int HR = (HP + 31) / 32;   // halo float4 per thread (8 per pixel)
  constexpr int ZQ = ZW / 4;           // float4 per dz row
  constexpr int ZRP = 256 / ZQ;        // dz rows per pass
  constexpr int ZR = BM / ZRP;         // dz float4 per thread
  __shared__ __attribute__((aligned(16))) float Hs[HP * HLD];
  __shared__ __attribute__((aligned(16))) float Zs[BM * ZLD];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int cchunk = blockIdx.y % nchunks;           // 32-channel input chunk
  const int co0 = (blockIdx.y / nchunks) * ZW;       // first output channel of this workgroup
  const int p_begin = blockIdx.x * patches_per_block;
  const int npatch = min(patches_per_block, total_patches - p_begin);
  if (npatch <= 0) return;

  const int hq = tid & 7, hr0 = tid >> 3;    // halo staging: float4 hq of halo pixel hr0 + 32 i
  const int zq = tid % ZQ, zr0 = tid / ZQ;   // dz staging: float4 zq of patch pixel zr0 + ZRP i
  f32x4 hreg[HR], zreg[ZR];
  auto load_patch = [&](int p) {
    const int tx = p % tiles_x, ty = (p / tiles_x) % tiles_y, img = p / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const float* ibase = d.in + (size_t)img * d.H * d.W * d.ldin + d.inoff + cchunk * 32 + hq * 4;
    const float* zbase = d.z + (size_t)img * d.H * d.W * d.ldz + d.zoff + co0 + zq * 4;
    // unconditional loads from a safe address + select (a branch around a load serialises the batch)
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hr0 + 32 * i;
      const int hy = hp / HWID, hx = hp - hy * HWID;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = hp < HP && y >= 0 && y < d.H && x >= 0 && x < d.W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? ibase + (size_t)(y * d.W + x) * d.ldin : d.in);
      hreg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
      const int pp = zr0 + ZRP * i;
      const int y = y0 + (pp >> 4), x = x0 + (pp & 15);
      const bool ok = y < d.H && x < d.W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? zbase + (size_t)(y * d.W + x) * d.ldz : d.z);
      zreg[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < HR; ++i)
      if (hr0 + 32 * i < HP) *reinterpret_cast<f32x4*>(Hs + (hr0 + 32 * i) * HLD + hq * 4) = hreg[i];
#pragma unroll
    for (int i = 0; i < ZR; ++i) *reinterpret_cast<f32x4*>(Zs + (zr0 + ZRP * i) * ZLD + zq * 4) = zreg[i];
  };

  // Static work split (no per-tile predicate inside the MFMA loop: a wave-uniform branch around every
  // ds_read + MFMA pair exposes the LDS latency each time).  Wavefront w owns output-channel tile
  // cot = w / PG and pixel-pair group pg = w % PG of the patch, and accumulates ALL 9 taps of that tile over
  // its pixels: 9 accumulator tiles = 144 registers; the PG partial sums meet in the final atomics.
  constexpr int PG = 4 / NCO;                 // pixel groups per co tile (NCO = 1: 4, NCO = 2: 2)
  constexpr int ROWS_PER_G = TH / PG;         // patch rows per pixel group
  const int cot = wave / PG, pg = wave % PG;
  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // lane part of the operand addresses: pixel pair member lh, column lr; group part: first patch row of the group
  const float* Zl = Zs + (pg * ROWS_PER_G * TW + lh) * ZLD + cot * 32 + lr;
  const float* Hl = Hs + (pg * ROWS_PER_G * HWID + lh) * HLD + lr;

  load_patch(p_begin);
  for (int pi = 0; pi < npatch; ++pi) {
    store_patch();
    __syncthreads();
    if (pi + 1 < npatch) load_patch(p_begin + pi + 1);
#pragma unroll
    for (int py = 0; py < ROWS_PER_G; ++py) {
#pragma unroll
      for (int tx2 = 0; tx2 < TW / 2; ++tx2) {   // pixel pair (row py of the group, columns 2*tx2 + lh)
        const float a = Zl[(py * TW + 2 * tx2) * ZLD];
        const float* hp = Hl + (py * HWID + 2 * tx2) * HLD;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float b = hp[((tap / 3) * HWID + (tap % 3)) * HLD];   // B operand: halo pixel shifted by the tap
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // Epilogue WITHOUT atomics: float atomics retire at only ~1.3 TB/s chip-wide (MI355X_MICROARCH.md) and every
  // workgroup ends with its whole tile set.  Each patch group (blockIdx.x) owns a private partial panel
  // dw + group * N * Kp; the PG pixel groups of a co tile first fold through LDS (staging buffers are free now),
  // then ONE wavefront per co tile writes plain 128-byte rows.  unpack_conv_grad sums the partial panels in a fixed
  // order, so the weight gradient is also bitwise reproducible.
  float* red = Hs;   // Hs and Zs are contiguous __shared__ arrays only by declaration order; use each separately
  float* part = d.dw + (size_t)blockIdx.x * d.N * d.Kp;
  constexpr int TILE = 16 * 64;   // floats of one accumulator tile across the wave
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    // fold the PG partial tiles of (cot, tap): groups 1..PG-1 publish, group 0 accumulates (one tap at a time keeps
    // the LDS footprint at NCO * (PG-1) * 4 KB)
    if (PG > 1) {
      __syncthreads();
      if (pg > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((cot * (PG - 1) + pg - 1) * 16 + r) * 64 + lane] = acc[tap][r];
      }
      __syncthreads();
      if (pg == 0) {
#pragma unroll
        for (int g2 = 0; g2 < PG - 1; ++g2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tap][r] += red[((cot * (PG - 1) + g2) * 16 + r) * 64 + lane];
      }
    }
    if (pg == 0) {
      // D[i = co][j = ci]: col = lane&31 -> ci, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> co
      float* base = part + (size_t)(co0 + cot * 32) * d.Kp + tap * d.Cp + cchunk * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        base[(size_t)row * d.Kp] = acc[tap][r];
      }
    }
  }
  (void)TILE;
}

template <int NCO>
static hipError_t launch_wg_halo(WgradDesc& d, hipStream_t s) {
  const int tiles_x = (d.W + 15) / 16, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B;
  const int nchunks = d.Cp / 32, ncog = d.N / (NCO * 32);
  // 2 workgroups per CU are resident; the grid is exactly that (every workgroup ends with 9*NCO tiles of float
  // atomics, which retire at only ~1.3 TB/s chip-wide, so fewer + longer workgroups beat finer load balance)
  int groups = (512 + nchunks * ncog - 1) / (nchunks * ncog);
  const size_t cap_groups = d.dw_capacity / ((size_t)d.N * d.Kp);   // one partial panel per patch group
  if ((size_t)groups > cap_groups) groups = (int)cap_groups;
  int ppb = (total + groups - 1) / groups;
  if (ppb < 4) ppb = 4;
  groups = (total + ppb - 1) / ppb;          // every group has >= 1 patch: every partial panel is fully written
  d.groups = groups;
  hipLaunchKernelGGL(wgrad3x3_halo_f32_kernel<NCO>, dim3(groups, nchunks * ncog), dim3(256), 0, s, d, tiles_x, tiles_y, total,
                     ppb, nchunks);
  return hipGetLastError();
}


hipError_t launch_wgrad_f32(WgradDesc& d, hipStream_t s) {
  d.groups = 1;   // every path sets the number of partial panels it wrote (the caller sums them: unpack_conv*_grad_kernel)
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return hipSuccess;
  if ((d.N & 3) || (d.ldz & 3) || (d.zoff & 3) || (d.Cp & 3) || (d.ldin & 3) || (d.inoff & 3) || d.K > d.Kp)
    return hipErrorInvalidValue;
  if (wgrad_thin_applicable(d)) return launch_wgrad_thin(d, s);       // first conv / 1x1 head: streaming kernels (wgrad_thin.hip)
  if (wino_wgrad_applicable(d)) return launch_wino_wgrad_f32(d, s);   // F(3x3,2x2): 2.25x fewer multiplies (wino_wgrad_f32.hip)
  if (tun(d).wgrad_halo && d.KS == 3 && d.Cp % 32 == 0 && d.N % 32 == 0 && d.K == 9 * d.Cp &&
      (long)d.H * d.W * d.ldin < (1l << 31) && (long)d.H * d.W * d.ldz < (1l << 31) &&
      d.dw_capacity >= (size_t)d.N * d.Kp) {
    if (d.N % 64 == 0) return launch_wg_halo<2>(d, s);
    return launch_wg_halo<1>(d, s);
  }
  if (d.dw_capacity < (size_t)d.N * d.Kp) return hipErrorInvalidValue;
  if (d.KS == 3) return launch_wg_tiles<3>(d, s);
  if (d.KS == 2) return launch_wg_tiles<2>(d, s);
  if (d.KS == 1) return launch_wg_tiles<1>(d, s);
  return hipErrorInvalidValue;
}

}  // namespace mgu
