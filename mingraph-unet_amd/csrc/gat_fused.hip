// Aggregate-FIRST graph attention for layers whose input is narrower than a head's output (Fin <= F'): the patch-graph
// GAT of the full forward (32 -> 4 heads x 64) and the C4 stress graph (64 -> 4 x 64).
//
// graph_attention.py:40-118 computes h = X W^T for every node, gathers h_src per edge and aggregates
// h'_j = sum_k alpha_k h_src(k).  The aggregate is LINEAR in h, so per head
//     h'_j = W_h ( sum_k alpha_k x_src(k) )
// i.e. the attention-weighted sum can be taken over the INPUT rows (Fin floats, shared by all heads) and the linear
// layer applied afterwards to the aggregated row.  For the patch GAT that replaces the gather of a 1 KiB row of Wh per
// edge by a 128-byte row of X (8x fewer gathered bytes), removes the (N, heads*F') node table (67 MB written + read at
// 64 graphs) and the GEMM launch that produced it; the attention logits need only s = X (W_h^T a_src), t = X (W_h^T a_tgt)
// (graph_attention.py:57-65), 2H scalars per node.  ELU is applied per head BEFORE the head mean (:118, :158), so the
// heads stay separate GEMMs of K = Fin.
//
//   gat_st_kernel     : st (N, 2H) = X [W^T a_src | W^T a_tgt]           (VALU, wave-uniform weights via the scalar cache)
//   gat_edge_max      : unchanged (gat.hip)
//   gat_fused_kernel  : one workgroup per 32 target nodes, one wavefront per head:
//        gather   -- Fin/4 lanes per node, each lane a 16-byte slice of x_src and the head's attention weight
//                    exp(LeakyReLU(s_src + t_tgt) - max_graph) (:86), accumulating 4 floats and the segment sum D;
//                    loads are unconditional (safe index + zero weight), two edges in flight per trip;
//        normalise-- agg_h = acc_h / (D_h + 1e-10) (:96) written to a per-wave LDS tile [H][32 nodes][Fin + 4];
//        linear   -- per head: D[node][f] = agg_h (32 x Fin) . W_h^T (Fin x F') on v_mfma_f32_32x32x2_f32, W_h^T
//                    fragments straight from L1/L2 in fragment order (pack_gat_wf_kernel);
//        epilogue -- ELU, concat store or head mean in registers; a lane owns one output channel (128-byte rows).
// The reference's result is reproduced up to fp32 reassociation (the sum over edges now happens before the dot
// products with W): the parity tests' 1e-3 bar is met with > 100x margin.
#include "common.h"
#include "gat_common.h"

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gf_dec_ordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// ---- st (N, 2H) = X WA^T, WA (2H, Fin) = [W_h^T a_src ; W_h^T a_tgt] ------------------------------------------------
template <int R2>   // 2H
__global__ __launch_bounds__(256) void gat_st_kernel(const float* __restrict__ x, const float* __restrict__ wa, int N, int Fin,
                                                     float* __restrict__ st) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool live = n < N;
  const float* xr = x + (size_t)(live ? n : 0) * Fin;
  float acc[R2];
#pragma unroll
  for (int r = 0; r < R2; ++r) acc[r] = 0.f;
#pragma unroll 1   // (unrolled, hipcc hoists every wave-uniform weight load and spills SGPRs)
  for (int c = 0; c < Fin; c += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
    for (int r = 0; r < R2; ++r) {
      const float* w = wa + r * Fin + c;   // wave-uniform: scalar loads
      acc[r] = fmaf(v[0], w[0], fmaf(v[1], w[1], fmaf(v[2], w[2], fmaf(v[3], w[3], acc[r]))));
    }
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < R2; ++r) st[(size_t)n * R2 + r] = acc[r];
  }
}

hipError_t launch_gat_st(const float* x, const float* wa, int N, int Fin, int heads, float* st, hipStream_t s) {
  if (N == 0) return hipSuccess;
  const dim3 grid((N + 255) / 256), block(256);
  switch (heads) {
    case 1: hipLaunchKernelGGL(gat_st_kernel<2>, grid, block, 0, s, x, wa, N, Fin, st); break;
    case 2: hipLaunchKernelGGL(gat_st_kernel<4>, grid, block, 0, s, x, wa, N, Fin, st); break;
    case 4: hipLaunchKernelGGL(gat_st_kernel<8>, grid, block, 0, s, x, wa, N, Fin, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- st AND the per-(graph, head) max of the attention logits in ONE launch -------------------------------------------
// The reference takes exp(e - max over ALL edges of the graph) (graph_attention.py:86).  The max needs s of an edge's SOURCE,
// which another workgroup may still be computing -- unless it is recomputed: s_i = (W_h^T a_src) . x_i is a Fin-long dot
// product of a row the gather brings in anyway.  Thread j computes s_j, t_j (written to st for the aggregate kernel) and, for
// each in-edge (i -> j), s_i again from x_i, e = LeakyReLU(s_i + t_j) (LeakyReLU is monotone: the max commutes with it), then
// wave-reduces and issues one order-encoded atomicMax per (graph, head).  That removes the separate gat_st and gat_edge_max
// launches and the dependent st[col[k]] round trip of the latter: a patch GAT layer is 2 launches instead of 4.
// The node -> graph id (binary search over graph_ptr through the scalar cache) is written to node_graph for the aggregate kernel.
__device__ __forceinline__ unsigned gf_enc_ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// FIN/4 lanes serve one node, each a 16-byte slice of the rows -- one 128- or 256-byte line per node and load instruction, as
// in the aggregate kernel (a row-per-lane version made the texture unit serve 64 different lines per instruction: 100 us at
// 64 graphs).  The partial dot products of a node's lanes are folded with DPP adds (quad permutes + row_shl; the sum lands in
// the node's first lanes), never through LDS permutes (a ds_bpermute fold: 30 us at 8 graphs); the weights of a lane's slice
// (2H x 4 floats) stay in registers; up to four in-edge rows are in flight per trip.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPN>
__device__ __forceinline__ float node_sum(float v) {   // valid in the first lanes of every LPN-lane group
  v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
  v = dpp_add<0x104>(v);         // row_shl:4  (lane i += lane i + 4)
  if (LPN == 16) v = dpp_add<0x108>(v);   // row_shl:8
  return v;
}

template <int FIN, int H>
__global__ __launch_bounds__(256) void gat_stmax_kernel(const float* __restrict__ x, const float* __restrict__ wa, int N,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const int32_t* __restrict__ gp, int G, float alpha, float* __restrict__ st,
                                                        int32_t* __restrict__ node_graph, gmax_t* __restrict__ gmax, int gstride, unsigned gen) {
  constexpr int LPN = FIN / 4;          // lanes per node
  constexpr int NPW = 64 / LPN;         // nodes per wavefront
  constexpr int EPT = 4;                // in-edges in flight per trip (the patch grid has <= 4)
  const int lane = threadIdx.x & 63;
  const int q = lane % LPN, ln = lane / LPN;
  const int n = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NPW + ln;
  if (n - ln >= N) return;   // wave-uniform
  const bool live = n < N;
  const int nd = live ? n : N - 1;
  f32x4 ws[H], wt[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    ws[h] = *reinterpret_cast<const f32x4*>(wa + h * FIN + 4 * q);
    wt[h] = *reinterpret_cast<const f32x4*>(wa + (H + h) * FIN + 4 * q);
  }
  auto dot = [](const f32x4 a, const f32x4 b) { return fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], a[3] * b[3]))); };
  const int k0 = rowptr[nd], deg = live ? rowptr[nd + 1] - k0 : 0;
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)nd * FIN + 4 * q);
  float s[H], t[H], m[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    s[h] = node_sum<LPN>(dot(xv, ws[h]));
    t[h] = node_sum<LPN>(dot(xv, wt[h]));
    m[h] = -INFINITY;
  }
  int g = 0;
  if (gp && G > 1) {
    int lo = 0, hi = G;  // gp[lo] <= node < gp[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (gp[mid] <= nd) lo = mid; else hi = mid;
    }
    g = lo;
  }
  if (live && q == 0) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      st[(size_t)n * (2 * H) + h] = s[h];
      st[(size_t)n * (2 * H) + H + h] = t[h];
    }
    if (node_graph) node_graph[n] = g;
  }
  // s of every in-neighbour, recomputed from its row; every load of a trip is issued before any is consumed (a missing edge
  // re-reads col[0]'s row and is masked afterwards: no branch around a load)
  const int maxdeg = (int)wave_max_u32((unsigned)deg);
  for (int e0 = 0; e0 < maxdeg; e0 += EPT) {
    f32x4 xs[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int j = col[e0 + u < deg ? k0 + e0 + u : 0];
      xs[u] = *reinterpret_cast<const f32x4*>(x + (size_t)j * FIN + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float sv = node_sum<LPN>(dot(xs[u], ws[h]));
        m[h] = e0 + u < deg ? fmaxf(m[h], sv) : m[h];
      }
  }
  const int g0 = __builtin_amdgcn_readfirstlane(g);
  const bool uniform = __all(g == g0 || !live);
  const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & (GMAX_SLOTS - 1);   // this wave's accumulator line (gat_common.h)
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float e = -INFINITY;
    if (q == 0 && deg > 0) {           // lane 0 of the node holds the folded sums
      e = m[h] + t[h];
      e = e > 0.f ? e : alpha * e;   // LeakyReLU is monotone: the max commutes with it
    }
    if (uniform) {
      e = wave_max_f32(e);
      if (lane == 0 && e > -INFINITY) gmax_add(gmax, gstride, slot, g0 * H + h, gen, e);
    } else if (e > -INFINITY) {
      gmax_add(gmax, gstride, slot, g * H + h, gen, e);
    }
  }
}

hipError_t launch_gat_stmax(const float* x, const float* wa, int N, int Fin, int heads, const int32_t* rowptr, const int32_t* col,
                            const int32_t* gp, int G, float alpha, float* st, int32_t* node_graph, unsigned long long* gmax, int gstride,
                            unsigned gen, hipStream_t s) {
  if (N == 0) return hipSuccess;
  const int npw = 64 / (Fin / 4);
  const dim3 grid((N + 4 * npw - 1) / (4 * npw)), block(256);
#define MGU_SM(FIN, H) hipLaunchKernelGGL((gat_stmax_kernel<FIN, H>), grid, block, 0, s, x, wa, N, rowptr, col, gp, G, alpha, st, node_graph, gmax, gstride, gen)
  if (Fin == 32 && heads == 1) MGU_SM(32, 1);
  else if (Fin == 32 && heads == 2) MGU_SM(32, 2);
  else if (Fin == 32 && heads == 4) MGU_SM(32, 4);
  else if (Fin == 64 && heads == 1) MGU_SM(64, 1);
  else if (Fin == 64 && heads == 2) MGU_SM(64, 2);
  else if (Fin == 64 && heads == 4) MGU_SM(64, 4);
  else return hipErrorInvalidValue;
#undef MGU_SM
  return hipGetLastError();
}

// ---- W_h^T in MFMA-fragment order: Wf[h][nt][kk][lane (kh = lane>>5, n = lane&31)][t] = W[h*Fh + 32 nt + n][8 kk + 4 kh + t]
__global__ void pack_gat_wf_kernel(const float* __restrict__ W, float* __restrict__ Wf, int heads, int Fh, int Fin) {
  const int total = heads * Fh * Fin;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i & 3, lane = (i >> 2) & 63;
    int rest = i >> 8;
    const int nkk = Fin / 8, nnt = Fh / 32;
    const int kk = rest % nkk;
    rest /= nkk;
    const int nt = rest % nnt, h = rest / nnt;
    Wf[i] = W[(size_t)(h * Fh + 32 * nt + (lane & 31)) * Fin + 8 * kk + 4 * (lane >> 5) + t];
  }
}

hipError_t launch_pack_gat_wf(const float* W, float* Wf, int heads, int Fh, int Fin, hipStream_t s) {
  const int total = heads * Fh * Fin;
  hipLaunchKernelGGL(pack_gat_wf_kernel, dim3((total + 255) / 256), dim3(256), 0, s, W, Wf, heads, Fh, Fin);
  return hipGetLastError();
}

// ---- everything that depends only on the layer's weights, in ONE launch (mgu_gat_prepare: once per weight version) ----
__device__ __forceinline__ int gf_graph_of(const int32_t* __restrict__ gp, int G, int node) {
  int lo = 0, hi = G;  // gp[lo] <= node < gp[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gp[mid] <= node) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void gat_prep_kernel(const float* __restrict__ W, const float* __restrict__ a, float* __restrict__ wa,
                                                       float* __restrict__ Wf, int heads, int Fh, int Fin, int nb_wf) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < 2 * heads) {
    // wa[r][k] = sum_f a[h][which*Fh + f] * W[h*Fh + f][k],  r = which*heads + h  (W_h^T a_src | W_h^T a_tgt)
    __shared__ float red[256];
    const int which = b / heads, h = b - which * heads;
    const int nfl = 256 / Fin, kl = t % Fin, fl = t / Fin;   // Fin in {32, 64}
    float sum = 0.f;
    if (fl < nfl)
      for (int f = fl; f < Fh; f += nfl) sum += a[h * 2 * Fh + which * Fh + f] * W[(size_t)(h * Fh + f) * Fin + kl];
    red[t] = sum;
    __syncthreads();
    if (fl == 0) {
      float tot = 0.f;
      for (int i = 0; i < nfl; ++i) tot += red[i * Fin + kl];
      wa[b * Fin + kl] = tot;
    }
  } else if (b < 2 * heads + nb_wf) {
    const int i = (b - 2 * heads) * 256 + t;
    if (i < heads * Fh * Fin) {
      const int tt = i & 3, lane = (i >> 2) & 63;
      int rest = i >> 8;
      const int nkk = Fin / 8, nnt = Fh / 32;
      const int kk = rest % nkk;
      rest /= nkk;
      const int nt = rest % nnt, h = rest / nnt;
      Wf[i] = W[(size_t)(h * Fh + 32 * nt + (lane & 31)) * Fin + 8 * kk + 4 * (lane >> 5) + tt];
    }
  }
}

hipError_t launch_gat_prep(const float* W, const float* a, float* wa, float* Wf, int heads, int Fh, int Fin, hipStream_t s) {
  const int nb_wf = (heads * Fh * Fin + 255) / 256;
  hipLaunchKernelGGL(gat_prep_kernel, dim3(2 * heads + nb_wf), dim3(256), 0, s, W, a, wa, Wf, heads, Fh, Fin, nb_wf);
  return hipGetLastError();
}

// One WORKGROUP per 32 target nodes, one WAVEFRONT per head (H waves): at the headline batch (8 graphs, 8192 nodes) there
// are only 256 node tiles, so a wave-per-tile kernel would leave 3/4 of every CU idle and run at single-wave latency;
// splitting the heads over waves gives H x the parallelism for the price of re-reading the (L1-resident) source rows.
template <int FIN, int NT, int H>
__global__ __launch_bounds__(64 * H) void gat_fused_kernel(const float* __restrict__ x, const float* __restrict__ st,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ node_graph,
                                                           const gmax_t* __restrict__ gmax, const float* __restrict__ Wf,
                                                           int N, int concat, float alpha, float* __restrict__ out,
                                                           int gstride, unsigned gen) {
  constexpr int LPN = FIN / 4;        // lanes per node in the gather (each a 16-byte slice of the input row)
  constexpr int NPP = 64 / LPN;       // nodes per gather pass
  constexpr int PASSES = 32 / NPP;
  constexpr int PLD = FIN + 4;        // LDS pitch of an aggregated row: conflict-free 16-byte reads, one node apart
  constexpr int P2 = 2 * H;
  constexpr int FO = 32 * NT;         // output width of a head
  constexpr int RLD = 32 + 4;         // pitch of the head-mean staging rows (one 32-channel n tile at a time)
  __shared__ __attribute__((aligned(16))) float agg_s[H][32 * PLD];
  __shared__ __attribute__((aligned(16))) float red_s[H][32 * RLD];
  const int lane = threadIdx.x & 63;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // this wave's head
  float* agg = agg_s[h];

  // XCD-aware order: workgroup b runs on XCD b % 8; give each XCD a contiguous range of node tiles so neighbouring
  // patch rows (j +- 1, j +- npw) are served by the same L2
  const int nblk = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, qn = nblk >> 3, rn = nblk & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
  const int n0 = tile * 32;

  // ---- gather + softmax-weighted aggregate of the INPUT rows (this head's weights) ---------------------------------
  // A lane serves PASSES nodes (node slot ln of every pass).  The dependent round trips of a node (rowptr -> col ->
  // source row) run for ALL its passes at once: PASSES independent chains in flight per lane.
  const int q = lane % LPN, ln = lane / LPN;
  int rs[PASSES], deg[PASSES];
  float ti[PASSES], gm[PASSES], D[PASSES];
  f32x4 acc[PASSES];
  int maxdeg = 0;
  // per-graph max of this head: one cooperative 64-slot read when the whole tile lies in one graph (the usual case)
  const int gt0 = node_graph ? node_graph[min(n0, N - 1)] : 0, gt1 = node_graph ? node_graph[min(n0 + 31, N - 1)] : 0;
  const bool one_graph = gt0 == gt1;                 // block-uniform
  const float gm_tile = one_graph ? gmax_read_wave(gmax, gstride, gt0 * H + h, gen) : 0.f;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int node = n0 + p * NPP + ln;
    const bool live = node < N;
    const int nd = live ? node : n0;
    rs[p] = rowptr[nd];
    deg[p] = live ? rowptr[nd + 1] - rs[p] : 0;
    maxdeg = max(maxdeg, deg[p]);
    const int g = node_graph ? node_graph[nd] : 0;
    ti[p] = st[(size_t)nd * P2 + H + h];
    gm[p] = one_graph ? gm_tile : gmax_read_lane(gmax, gstride, g * H + h, gen);
    D[p] = 0.f;
    acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int EPT = FIN == 32 ? 4 : 2;   // edges per trip (in-degree of the patch grid is <= 4: one trip)
  for (int e0 = 0; __any(e0 < maxdeg); e0 += EPT) {
    f32x4 xj[EPT][PASSES];
    float sj[EPT][PASSES];
    // every load of the trip is issued before any is consumed; a missing edge reads a mapped address (col[0] -> some
    // row) and gets weight 0 -- no branch around a load
#pragma unroll
    for (int u = 0; u < EPT; ++u)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int j = col[e0 + u < deg[p] ? rs[p] + e0 + u : 0];
        xj[u][p] = *reinterpret_cast<const f32x4*>(x + (size_t)j * FIN + 4 * q);
        sj[u][p] = st[(size_t)j * P2 + h];
      }
#pragma unroll
    for (int u = 0; u < EPT; ++u)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float ev = sj[u][p] + ti[p];
        ev = ev > 0.f ? ev : alpha * ev;                              // LeakyReLU (:65)
        const float w = e0 + u < deg[p] ? __expf(ev - gm[p]) : 0.f;   // exp(e - max(e)) (:86)
        D[p] += w;
        acc[p] += w * xj[u][p];
      }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const float inv = 1.f / (D[p] + 1e-10f);                          // (:96); a node without in-edges aggregates to 0
    *reinterpret_cast<f32x4*>(agg + (p * NPP + ln) * PLD + 4 * q) = acc[p] * inv;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the aggregated tile is private to this wave: LDS completion only
  __builtin_amdgcn_wave_barrier();

  // ---- (32 nodes x FIN) . W_h^T (FIN x 32 NT) on the fp32 matrix cores, ELU, concat | mean --------------------------
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 c[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[nt][r] = 0.f;
  f32x4 bw[NT][FIN / 8];   // W_h^T fragments from L1/L2, all requested up front
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int kk = 0; kk < FIN / 8; ++kk)
      bw[nt][kk] = *reinterpret_cast<const f32x4*>(Wf + ((size_t)((h * NT + nt) * (FIN / 8) + kk) * 64 + lane) * 4);
#pragma unroll
  for (int kk = 0; kk < FIN / 8; ++kk) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(agg + lr * PLD + kk * 8 + lh * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) c[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bw[nt][kk][t], c[nt], 0, 0, 0);
  }
  const float invH = 1.f / H;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = c[nt][r];
      v = v > 0.f ? v : (__expf(v) - 1.f);                            // ELU (:118)
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (concat) {
        if (n0 + row < N) out[(size_t)(n0 + row) * (H * FO) + h * FO + nt * 32 + lr] = v;   // cat over heads (:155)
      } else {
        red_s[h][row * RLD + lr] = v * invH;
      }
    }
    if (!concat) {   // mean over heads (:158): the H waves meet in LDS (one n tile at a time), then 16-byte row stores
      __syncthreads();
      for (int u = threadIdx.x; u < 32 * 8; u += 64 * H) {
        const int row = u >> 3, qd = u & 7;
        f32x4 sum = *reinterpret_cast<const f32x4*>(&red_s[0][row * RLD + 4 * qd]);
#pragma unroll
        for (int hh = 1; hh < H; ++hh) sum += *reinterpret_cast<const f32x4*>(&red_s[hh][row * RLD + 4 * qd]);
        if (n0 + row < N) *reinterpret_cast<f32x4*>(out + (size_t)(n0 + row) * FO + nt * 32 + 4 * qd) = sum;
      }
      if (nt + 1 < NT) __syncthreads();
    }
  }
}

bool gat_fused_applicable(int Fin, int heads, int Fh, int64_t E) {
  return (Fin == 32 || Fin == 64) && (Fh == 32 || Fh == 64) && (heads == 1 || heads == 2 || heads == 4) && Fin <= Fh && E > 0;
}

size_t gat_fused_scratch_floats(int Fin, int heads, int Fh) { return (size_t)heads * Fh * Fin + (size_t)2 * heads * Fin + 64; }

template <int FIN, int NT, int H>
static hipError_t launch_fused_t(const float* x, const float* st, const int32_t* rowptr, const int32_t* col,
                                 const int32_t* node_graph, const unsigned long long* gmax, const float* Wf, int N, int concat, float alpha,
                                 float* out, int gstride, unsigned gen, hipStream_t s) {
  const int ntiles = (N + 31) / 32;
  hipLaunchKernelGGL((gat_fused_kernel<FIN, NT, H>), dim3(ntiles), dim3(64 * H), 0, s, x, st, rowptr, col, node_graph, gmax, Wf, N,
                     concat, alpha, out, gstride, gen);
  return hipGetLastError();
}

hipError_t launch_gat_fused(const float* x, int Fin, const float* st, const int32_t* rowptr, const int32_t* col,
                            const int32_t* node_graph, const unsigned long long* gmax, const float* Wf, int N, int heads, int Fh, int concat,
                            float alpha, float* out, int gstride, unsigned gen, hipStream_t s) {
  if (N == 0) return hipSuccess;
#define MGU_GF(FIN, NT, H) \
  return launch_fused_t<FIN, NT, H>(x, st, rowptr, col, node_graph, gmax, Wf, N, concat, alpha, out, gstride, gen, s)
#define MGU_GF_H(FIN, NT)      \
  do {                         \
    if (heads == 1) MGU_GF(FIN, NT, 1); \
    if (heads == 2) MGU_GF(FIN, NT, 2); \
    if (heads == 4) MGU_GF(FIN, NT, 4); \
  } while (0)
  if (Fin == 32 && Fh == 32) MGU_GF_H(32, 1);
  if (Fin == 32 && Fh == 64) MGU_GF_H(32, 2);
  if (Fin == 64 && Fh == 64) MGU_GF_H(64, 2);
#undef MGU_GF_H
#undef MGU_GF
  return hipErrorInvalidValue;
}

}  // namespace mgu
