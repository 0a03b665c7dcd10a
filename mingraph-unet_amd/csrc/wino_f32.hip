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
//   * per 32 input channels the RAW 10 x 34 halo is staged once into LDS -- the transformed input V is never
//     written anywhere.  Wavefront (i, g) owns row i of the 4x4 transform: row i of B^T d is a +-1 combination of
//     two raw rows, so the wave reads 2 rows x 4 columns (8 ds_read_b128) per lane and k chunk, forms its 4
//     components V[i][0..3] with 8 vector adds, and issues 16 MFMAs on them.  That is 2 LDS reads per V value --
//     fewer than writing V to LDS and reading it back -- with no barrier between transform and MFMA;
//   * raw columns are stored split by parity ([row][x & 1][x >> 1][36 floats]): tile tx touches entries tx, tx+1 of
//     each parity plane, so the 16 lanes of a ds_read_b128 group are 144 bytes apart -> conflict free;
//   * U = G g G^T is precomputed (pack_wino_w_kernel) in MFMA-fragment order: the four components of a wave are one
//     contiguous 4 KB block per 8 input channels, loaded straight from L2 into VGPRs (coalesced 16-byte lanes).
//     Only one wave of the workgroup uses a given component, so staging U through LDS would buy nothing;
//   * inverse transform: each wave folds its own row (M[i][.] A) in registers, the four row waves meet through a
//     small LDS exchange, and lanes store one channel each (32 lanes = one 128-byte line of a pixel).
#include "common.h"

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static bool g_use_wino = true;   // MGU_NO_WINOGRAD=1: direct (halo implicit-GEMM) kernels only
void set_use_wino(bool on) { g_use_wino = on; }
bool use_wino() { return g_use_wino; }

// U[ntile][cin/8][i*4+j][lane (h = lane>>5, r = lane&31)][t]  =  (G g G^T)[i][j]  of  cout = 32*ntile + r,
// cin = 8*(cin/8) + 4*h + t.   dgrad = 1: the data-gradient conv, g'[u][v] = w[c][n][2-u][2-v] (roles swapped).
__global__ void pack_wino_w_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Cp, int Np,
                                   int dgrad) {
  const int64_t total = (int64_t)Np * Cp;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
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
    float* dst = U + (((int64_t)(n >> 5) * (Cp >> 3) + (c >> 3)) * 16) * 256 + ((((c >> 2) & 1) * 32 + (n & 31)) * 4 + (c & 3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dst[(i * 4 + 0) * 256] = t[i][0];
      dst[(i * 4 + 1) * 256] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      dst[(i * 4 + 2) * 256] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      dst[(i * 4 + 3) * 256] = t[i][2];
    }
  }
}

size_t wino_u_floats(int Cout, int Cp) { return (size_t)((Cout + 63) / 64 * 64) * Cp * 16; }   // n tiles padded to pairs

hipError_t launch_pack_wino_w(const float* w, float* U, int Cout, int Cin, int Cp, int dgrad, hipStream_t s) {
  if (Cp & 7) return hipErrorInvalidValue;
  const int Np = (Cout + 63) / 64 * 64;
  int64_t blocks = ((int64_t)Np * Cp + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(pack_wino_w_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, U, Cout, Cin, Cp, Np, dgrad);
  return hipGetLastError();
}

// NT = 2: 64 output channels per workgroup, wave (i, g) owns n tile g and both m tiles;
// NT = 1: 32 output channels,               wave (i, g) owns m tile g.
template <int NT>
__global__ __launch_bounds__(512) void wino3x3_f32_kernel(const IgemmDesc d, const int tiles_x, const int tiles_y,
                                                          const int total_patches, const int patches_per_block,
                                                          const int ngroups, const int nitems, const int per_xcd) {
  constexpr int MT = NT;                         // m tiles per wavefront
  constexpr int RH = 10, RW = 34, HPIX = RH * RW;   // raw halo of the 8 x 32 pixel patch
  constexpr int PLD = 36;                        // floats per raw pixel in LDS: 32 channels + 4 pad (144 bytes)
  constexpr int HR = (HPIX + 63) / 64;           // raw pixels staged per thread (8 threads x 16 bytes per pixel)
  constexpr int S1 = 17 * PLD, S2 = PLD, S3 = 17 * PLD + PLD;   // LDS offsets of tile columns 1..3 (parity planes)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                 // [RH][2][17][PLD]
  float* Zx = smem + HPIX * PLD;    // [4 rows i][2 g][MT][16 regs][64 lanes]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, wg = wave >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  const int tx = lr & 15, ty = lr >> 4;
  // XCD-aware work order: workgroup L runs on XCD L % 8 (round-robin dispatch); give each XCD a CONTIGUOUS range of
  // (n block, patch group) items in n-major order, so the workgroups that share an L2 stream the same U slice (large
  // layers: 2-4 MB per n block against a 4 MB L2) and neighbouring patches (shared halo rows).
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
    const int m_abs = NT == 2 ? mi : wg;
    const int rowbase = 2 * (2 * m_abs + ty);
    offA[mi] = ((rowbase + ra) * 34 + tx) * PLD + lh * 4;
    offB[mi] = ((rowbase + rb) * 34 + tx) * PLD + lh * 4;
  }
  const int ncg = d.Cp >> 3;                       // 8-channel k groups
  const int nC = d.Cp >> 5;                        // 32-channel raw chunks
  const int ntg = nblock * NT + (NT == 2 ? wg : 0);
  const float* const up = d.wu + ((size_t)ntg * ncg * 16 + wi * 4) * 256 + lane * 4;

  // ---- raw halo staging: thread -> (pixel hp0 + 64 i, 16-byte piece kq) ----
  const int kq = tid & 7, hp0 = tid >> 3;
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
      const int hp = hp0 + 64 * i;
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
    for (int i = 0; i < HR; ++i) hreg[i] = *reinterpret_cast<const f32x4*>(load_base + hoff[i] + c * 32);
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + 64 * i;
      const int r = hp / RW, cc = hp - r * RW;
      if (hp < HPIX)
        *reinterpret_cast<f32x4*>(Hs + ((r * 2 + (cc & 1)) * 17 + (cc >> 1)) * PLD + kq * 4) =
            ((hmask >> i) & 1u) ? hreg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 bf[2][4];
  auto load_b = [&](int cg, int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[slot][j] = *reinterpret_cast<const f32x4*>(up + (size_t)cg * 4096 + j * 256);
  };

  f32x16 acc[4][MT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][mi][r] = 0.f;

  setup_load(p_begin);
  load_halo(0);
  load_b(0, 0);
  int cg = 0;   // k group (of the patch) whose fragments sit in bf[0] at the top of a chunk
  for (int pi = 0; pi < npatch; ++pi) {
    for (int c = 0; c < nC; ++c) {
      __syncthreads();   // every wave is done with the previous raw chunk
      store_halo();
      __syncthreads();   // raw chunk visible
      if (c + 1 < nC) {
        load_halo(c + 1);
      } else if (pi + 1 < npatch) {
        setup_load(p_begin + pi + 1);
        load_halo(0);
      }
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        int nx = cg + 1;
        if (nx == ncg) nx = 0;
        load_b(nx, (kg + 1) & 1);   // fragments of the next k group (wraps to the next patch's first)
        __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: hipcc otherwise sinks them to just before their use
        cg = nx;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          f32x4 v[4];
          {
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
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[j][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][t], bf[kg & 1][j][t], acc[j][mi], 0, 0, 0);
        }
      }
    }
    // ---- inverse transform + epilogue of patch pi --------------------------------------------------------
    int img, y0, x0;
    setup_patch(p_begin + pi, img, y0, x0);
    float* const img_out = d.out + (size_t)img * d.H * d.W * d.ldout + d.coff;
    const int n = ntg * 32 + lr;
    const bool nvalid = n < d.N;
    const float sc = (nvalid && d.scale) ? d.scale[n] : 1.f;
    const float sh = (nvalid && d.shift) ? d.shift[n] : 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // column part in registers: Z[i][q] = sum_j M[i][j] A[j][q]   (A^T = [1 1 1 0; 0 1 -1 -1])
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = q == 0 ? (acc[0][mi][r] + acc[1][mi][r] + acc[2][mi][r]) : (acc[1][mi][r] - acc[2][mi][r] - acc[3][mi][r]);
          Zx[(((wi * 2 + wg) * MT + mi) * 16 + r) * 64 + lane] = z;
        }
      __syncthreads();
      // row part: wave i finishes accumulator registers 4i .. 4i+3 (tile row 2m + (i>>1), tiles 8(i&1) + 4h + 0..3)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int m_abs = NT == 2 ? mi : wg;
        const int oy = y0 + 2 * (2 * m_abs + (wi >> 1));
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 4 * wi + rr;
          const float z0 = Zx[(((0 * 2 + wg) * MT + mi) * 16 + r) * 64 + lane];
          const float z1 = Zx[(((1 * 2 + wg) * MT + mi) * 16 + r) * 64 + lane];
          const float z2 = Zx[(((2 * 2 + wg) * MT + mi) * 16 + r) * 64 + lane];
          const float z3 = Zx[(((3 * 2 + wg) * MT + mi) * 16 + r) * 64 + lane];
          float ya = (z0 + z1 + z2) * sc + sh;    // output row 2*tr
          float yb = (z1 - z2 - z3) * sc + sh;    // output row 2*tr + 1
          if (d.relu) ya = fmaxf(ya, 0.f), yb = fmaxf(yb, 0.f);
          const int ox = x0 + 2 * (rr + 8 * (wi & 1) + 4 * lh) + q;
          if (nvalid && ox < d.W) {
            if (oy < d.H) img_out[(size_t)(oy * d.W + ox) * d.ldout + n] = ya;
            if (oy + 1 < d.H) img_out[(size_t)((oy + 1) * d.W + ox) * d.ldout + n] = yb;
          }
        }
      }
      __syncthreads();   // Zx is rewritten by the next pass / patch
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][mi][r] = 0.f;
  }
}

template <int NT>
static hipError_t launch_wino_nt(const IgemmDesc& d, hipStream_t s) {
  const int tiles_x = (d.W + 31) / 32, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B, nblk = (d.N + 32 * NT - 1) / (32 * NT);
  // one 512-thread workgroup per CU is resident: keep ~2 workgroups per CU in the grid, each walking its patches
  int ppb = (int)(((long)total * nblk) / (256 * 2));
  if (ppb < 1) ppb = 1;
  if (ppb > 16) ppb = 16;
  const int ngroups = (total + ppb - 1) / ppb;
  const int per_xcd = (ngroups * nblk + 7) / 8;
  dim3 grid(8 * per_xcd, 1);
  const size_t lds = (size_t)(10 * 34 * 36 + 4 * 2 * NT * 16 * 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino3x3_f32_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(wino3x3_f32_kernel<NT>, grid, dim3(512), lds, s, d, tiles_x, tiles_y, total, ppb, ngroups,
                     ngroups * nblk, per_xcd);
  return hipGetLastError();
}

bool wino_applicable(const IgemmDesc& d) {
  return g_use_wino && d.wu && d.KS == 3 && d.out_mode == 0 && d.split_n == 0 && (d.Cp % 32) == 0 && d.K == 9 * d.Cp &&
         (d.ldin & 3) == 0 && (long)d.H * d.W * d.ldin < (1l << 31) && (long)d.H * d.W * d.ldout < (1l << 31);
}

hipError_t launch_wino_f32(const IgemmDesc& d, hipStream_t s) {
  if (d.N > 32) return launch_wino_nt<2>(d, s);
  return launch_wino_nt<1>(d, s);
}

}  // namespace mgu
