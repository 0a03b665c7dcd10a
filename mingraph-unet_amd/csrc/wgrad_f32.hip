// Weight-gradient GEMM for gfx950, fp32 MFMA: Dw[n][k] += sum_m Z[m][n] * A(m,k).
//
// Backward of nn.Conv2d / nn.ConvTranspose2d w.r.t. the weight (the autograd node behind
// loss.backward() at scripts/train_segmentation.py:133).  Z is the gradient flowing into the
// convolution's output (rows = output pixels), A(m,k) is the same on-the-fly im2col gather the forward
// kernel uses (k = tap*Cp + c), so no column buffer is ever materialised.  The reduction runs over the
// PIXEL index m, which is the slow (row) index of both operands in NHWC memory; with the 32x32x2 fp32
// MFMA a lane supplies ONE float per operand (A[i = lane&31][kk = lane>>5]), so operands are fetched
// from m-major LDS tiles with conflict-free ds_read_b32 -- no transpose anywhere.
// The pixel range is split over blockIdx.y; partial tiles are combined with no-return
// global_atomic_add_f32 (each wave instruction adds two contiguous 128-byte row segments, the
// full-rate shape in MI355X_MICROARCH.md "Global float atomics").
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

#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WKT; ++j) {
      const int kk = k0 + (wk * WKT + j) * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * WNT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < d.N && kk < d.K) atomicAdd(d.dw + (size_t)n * d.Kp + kk, acc[i][j][r]);
      }
    }
}

template <int KS, int WAVES_N, int WAVES_K, int WNT, int WKT>
static hipError_t launch_wg(const WgradDesc& d0, hipStream_t s) {
  constexpr int BN = WAVES_N * WNT * 32;
  WgradDesc d = d0;
  const int ntiles = (d.N + BN - 1) / BN, ktiles = (d.K + WG_BK - 1) / WG_BK;
  // split the pixel range so the grid has ~4 workgroups per CU, >= 8 steps per split
  int splits = (1024 + ntiles * ktiles - 1) / (ntiles * ktiles);
  int rows = (d.M + splits - 1) / splits;
  if (rows < 8 * WG_MK) rows = 8 * WG_MK;
  rows = (rows + WG_MK - 1) / WG_MK * WG_MK;
  splits = (d.M + rows - 1) / rows;
  d.rows_per_split = rows;
  hipLaunchKernelGGL((wgrad_f32_kernel<KS, WAVES_N, WAVES_K, WNT, WKT>), dim3(ntiles * ktiles, splits), dim3(256), 0, s, d);
  return hipGetLastError();
}

template <int KS>
static hipError_t launch_wg_tiles(const WgradDesc& d, hipStream_t s) {
  if (d.N > 64) return launch_wg<KS, 2, 2, 2, 2>(d, s);   // 128 (n) x 128 (k), wave 64x64
  if (d.N > 32) return launch_wg<KS, 2, 2, 1, 2>(d, s);   // 64 x 128, wave 32x64
  return launch_wg<KS, 1, 4, 1, 1>(d, s);                 // 32 x 128, wave 32x32
}

hipError_t launch_wgrad_f32(const WgradDesc& d, hipStream_t s) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return hipSuccess;
  if ((d.N & 3) || (d.ldz & 3) || (d.zoff & 3) || (d.Cp & 3) || (d.ldin & 3) || (d.inoff & 3) || d.K > d.Kp)
    return hipErrorInvalidValue;
  if (d.KS == 3) return launch_wg_tiles<3>(d, s);
  if (d.KS == 2) return launch_wg_tiles<2>(d, s);
  if (d.KS == 1) return launch_wg_tiles<1>(d, s);
  return hipErrorInvalidValue;
}

}  // namespace mgu
