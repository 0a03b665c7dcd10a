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
//   gat_stmax_kernel  : st (N, 2H) = X [W^T a_src | W^T a_tgt] and the per-graph maxima of the attention logits, one launch
//   gat_fused2_kernel : gather + softmax-weighted aggregate of the INPUT rows for all heads, the per-head linear layer on the bf16
//                       matrix cores with exact three-way operand splits, ELU, concat store or head mean (see the kernel's comment)
// The reference's result is reproduced up to fp32 reassociation (the sum over edges now happens before the dot
// products with W): the parity tests' 1e-3 bar is met with > 100x margin.
#include <algorithm>

#include "common.h"
#include <atomic>
#include "gat_common.h"
#include "x3.h"

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gf_dec_ordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
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

__device__ __forceinline__ int gf_graph_of(const int32_t* __restrict__ gp, int G, int node) {
  int lo = 0, hi = G;  // gp[lo] <= node < gp[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gp[mid] <= node) lo = mid; else hi = mid;
  }
  return lo;
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
  const bool live = n < N;   // (a wave past the end of the node range runs on, clamped to the last node, without stores: it takes part in
                             // the workgroup barrier in front of the accumulator atomics)
  const int nd = live ? n : N - 1;
  f32x4 ws[H], wt[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    ws[h] = *reinterpret_cast<const f32x4*>(wa + h * FIN + 4 * q);
    wt[h] = *reinterpret_cast<const f32x4*>(wa + (H + h) * FIN + 4 * q);
  }
  auto dot = [](const f32x4 a, const f32x4 b) { return fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], a[3] * b[3]))); };
  // (the degree through a mask, not a select of the second load: hipcc sinks a load that is needed on one side of a select under
  // the condition, and the branch carries a vmcnt(0))
  const int k0 = rowptr[nd], deg = (rowptr[nd + 1] - k0) & (live ? -1 : 0);
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)nd * FIN + 4 * q);
  float s[H], t[H], m[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    s[h] = node_sum<LPN>(dot(xv, ws[h]));
    t[h] = node_sum<LPN>(dot(xv, wt[h]));
    m[h] = -INFINITY;
  }
  // node -> graph: the wave's nodes are consecutive, so the search runs ONCE per wave on the scalar unit (wave-uniform index:
  // s_load through the scalar cache) for its first and last node; only a wave that straddles a graph boundary searches per lane
  // (a per-lane binary search is log2(G) DEPENDENT vector loads: six round trips in front of everything at 64 graphs)
  int g = 0;
  if (gp && G > 1) {
    const int n_first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NPW;
    const int g_first = gf_graph_of(gp, G, min(n_first, N - 1));
    g = g_first;
    if (min(n_first + NPW - 1, N - 1) >= gp[g_first + 1]) g = gf_graph_of(gp, G, nd);   // wave-uniform condition
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
  // The accumulator atomics execute at the memory side, ~50 ns each per CU whatever the address (MI355X_MICROARCH.md "Global float
  // atomics"): with one per (wave, head) they, not the arithmetic, were what the kernel's time grew with (128 per CU at 64 graphs).
  // The four waves of a workgroup meet in LDS first: one atomic per (workgroup, head) when they sit in one graph (the usual case).
  __shared__ float wmax_s[4][H];
  __shared__ int wgraph_s[4];
  const int wv = threadIdx.x >> 6;
  float ew[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float e = -INFINITY;
    if (q == 0 && deg > 0) {           // lane 0 of the node holds the folded sums
      e = m[h] + t[h];
      e = e > 0.f ? e : alpha * e;   // LeakyReLU is monotone: the max commutes with it
    }
    if (uniform) e = wave_max_f32(e);
    ew[h] = e;
  }
  if (lane == 0) {
    wgraph_s[wv] = uniform ? g0 : -1;
#pragma unroll
    for (int h = 0; h < H; ++h) wmax_s[wv][h] = ew[h];
  }
  constexpr int nwaves = 4;
  __syncthreads();
  bool merged = uniform;
  for (int w2 = 0; w2 < nwaves; ++w2) merged = merged && wgraph_s[w2] == g0;   // block-uniform: every wave uniform, one graph
  if (merged) {
    if (wv == 0 && lane < H) {
      float e = wmax_s[0][lane];
      for (int w2 = 1; w2 < nwaves; ++w2) e = fmaxf(e, wmax_s[w2][lane]);
      if (e > -INFINITY) gmax_add(gmax, gstride, slot, g0 * H + lane, gen, e);
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (uniform) {
        if (lane == 0 && ew[h] > -INFINITY) gmax_add(gmax, gstride, slot, g0 * H + h, gen, ew[h]);
      } else if (ew[h] > -INFINITY) {
        gmax_add(gmax, gstride, slot, g * H + h, gen, ew[h]);
      }
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

// ---- everything that depends only on the layer's weights, in ONE launch (mgu_gat_prepare: once per weight version) ----
__global__ __launch_bounds__(256) void gat_prep_kernel(const float* __restrict__ W, const float* __restrict__ a, float* __restrict__ wa,
                                                       unsigned* __restrict__ Wx, int heads, int Fh, int Fin, int nb_wx) {
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
  } else if (b < 2 * heads + nb_wx) {
    // Wx[item = h * NT + nt][ks][piece][lane (hk = lane >> 5, n = lane & 31)][8 bf16]: the B fragment of v_mfma_f32_32x32x16_bf16 of
    // W[h * Fh + 32 nt + n][16 ks + 8 hk + e], split exactly into three bf16 pieces (x3.h); one thread per (item, ks, lane)
    const int i = (b - 2 * heads) * 256 + t;
    const int nks = Fin / 16, nnt = Fh / 32;
    if (i < heads * nnt * nks * 64) {
      const int lane = i & 63, ks = (i >> 6) % nks, item = (i >> 6) / nks;
      const int h = item / nnt, nt = item % nnt;
      const float* wr = W + (size_t)(h * Fh + 32 * nt + (lane & 31)) * Fin + 16 * ks + 8 * (lane >> 5);
      u32x4 p0, p1, p2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned q0, q1, q2;
        split3_pack(wr[2 * e], wr[2 * e + 1], q0, q1, q2);
        p0[e] = q0, p1[e] = q1, p2[e] = q2;
      }
      u32x4* dst = reinterpret_cast<u32x4*>(Wx) + ((size_t)(item * nks + ks) * 3) * 64 + lane;
      dst[0] = p0, dst[64] = p1, dst[128] = p2;
    }
  }
}

// floats of the prepared-weight buffer: [wa (2H, Fin) | Wx (three bf16 pieces of W_h^T in fragment order: 6 bytes per weight)]
size_t gat_fused_scratch_floats(int Fin, int heads, int Fh) { return (size_t)2 * heads * Fin + (size_t)heads * Fh * Fin * 3 / 2 + 64; }

hipError_t launch_gat_prep(const float* W, const float* a, float* wa, unsigned* Wx, int heads, int Fh, int Fin, hipStream_t s) {
  const int nb_wx = (heads * (Fh / 32) * (Fin / 16) * 64 + 255) / 256;
  hipLaunchKernelGGL(gat_prep_kernel, dim3(2 * heads + nb_wx), dim3(256), 0, s, W, a, wa, Wx, heads, Fh, Fin, nb_wx);
  return hipGetLastError();
}

// =====================================================================================================================
// gat_fused2_kernel (round 4) -- the same layer schedule with the redundancy of gat_fused_kernel removed:
//   * ONE gather of a source row per edge for ALL heads (gat_fused_kernel gave every head its own wave, and each wave re-read the
//     rows and the CSR: H x the load instructions);
//   * ONE softmax weight per (edge, head): lane q of a node's lane group evaluates slot q of the trip -- exp(LeakyReLU(s_src + t_tgt)
//     - max_graph) for the H heads -- and the group shares the weights (and the source ids) through an LDS line; gat_fused_kernel
//     evaluated every weight in all FIN/4 lanes of the group;
//   * the linear layer on v_mfma_f32_32x32x16_bf16 with the exact three-way bf16 operand split of the Winograd kernels (x3.h): an
//     fp32 GEMM in accuracy, 6 x 32 cycles per 16 input features instead of 8 x 64 on the fp32 MFMA.  W_h^T pieces are packed in
//     fragment order by gat_prep_kernel and stay in REGISTERS across the tiles of a persistent workgroup;
//   * persistent workgroups (a few per CU) walking node tiles, so the weight pieces, the per-graph maxima and the launch overhead
//     are paid once per workgroup, not once per 32 nodes.
// Dependent global round trips per tile: rowptr -> col -> {x row, s} (three; the rows and the attention scalars travel together).
// Workgroup = 4 wavefronts on a 32-node tile: gather by (node, 16-byte row slice) lanes, then the (head, n tile) GEMM items are
// dealt to the waves, then ELU and the concat store or the head mean through LDS (fixed order: bitwise reproducible).
// Workgroup barrier of gat_fused2_kernel.  With this toolchain __syncthreads() is `s_waitcnt lgkmcnt(0); s_barrier` -- the workgroup-
// scope fence no longer drains vmcnt, so the prefetched rows of the next tile and the output stores of this one stay in flight across
// it.  (An inline-asm barrier of the same two instructions gave the GEMM behind barrier A a different instruction order, and THAT
// order produced one wrong row in about every fourth call on the configs[3] graphs -- single rows off by 1e-4 .. 1e-3, never with
// this form in 120 calls; a read-back of the aggregate tile in front of the barrier did not cure it, so the hand-off itself is not
// what failed.  tests/test_gpu_gat_schedules.py::test_repeated_calls_are_bitwise_identical repeats the layer 40 times.)
// Round 5, from the ISA of both forms built from this source (hipcc -S, kernel <64, 2, 4>; tools/ not needed -- `grep -n s_barrier`):
// the two builds have the same three barriers per tile, the same ds_write / ds_read sets on each side of every barrier (no LDS access
// of the aggregate tile, the weight lines or the exchange rows sits on the wrong side in either build), the same counted lgkmcnt
// waits in front of every LDS-fed MFMA, and no MFMA result is read by a VALU or LDS instruction inside the MFMA -> VALU hazard
// window; what differs is only the interleaving of the GEMM's ds_read_b128 pairs with its MFMAs (the asm form issues two MFMAs
// before the third read pair, this form one).  So the wrong row of round 4 is NOT explained by an access crossing the barrier;
// what this form has and the asm form lacked is the workgroup-scope fence pair around s_barrier (__syncthreads() = fence,
// s_barrier, fence): the asm statement's "memory" clobber orders only the compiler's view of memory, while the fences also pin
// the backend's waitcnt bookkeeping to the barrier.  The hand-off is therefore kept on the fenced form BY CONSTRUCTION, not by
// measurement; the cause of the round-4 failure at the ISA level remains unidentified (recorded as such in NOTES.md).
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }

#if defined(MGU_DIAG) && (MGU_DIAG == 40 || MGU_DIAG == 41)
// diagnostic builds (never shipped): phase stamps of one workgroup of gat_fused2_kernel, mgu_diag_gat[wave][tile][slot].
//   40: every stamp drains the wave's memory operations first, so the intervals are the LATENCIES of the phases (and of the stamp's
//       own store: ~1000-2000 cycles of floor per interval);
//   41: no drain -- the stamps go to LDS and are copied out when the workgroup is done: the pipelined cost of the phases.
__device__ unsigned long long mgu_diag_gat[4][8][16];
#if MGU_DIAG == 40
#define GAT_T(slot)                                                                                                    \
  do {                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                         \
    if (blockIdx.x == 100 && lane == 0 && dti < 8) mgu_diag_gat[wave][dti][slot] = __builtin_readcyclecounter();       \
  } while (0)
#else
#define GAT_T(slot)                                                                                \
  do {                                                                                             \
    if (lane == 0 && dti >= 0 && dti < 8) diag_ts[wave][dti][slot] = __builtin_readcyclecounter(); \
  } while (0)
#endif
#else
#define GAT_T(slot) do {} while (0)
#endif
template <int FIN, int NT, int H>
__global__ __launch_bounds__(256, FIN == 32 ? 3 : 2) void gat_fused2_kernel(const float* __restrict__ x, const float* __restrict__ st,
                                                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ gp, int G, const gmax_t* __restrict__ gmax,
                                                         const unsigned* __restrict__ Wx, int N, int ntiles, int concat, float alpha,
                                                         float* __restrict__ out, int gstride, unsigned gen) {
  constexpr int LPN = FIN / 4;        // lanes per node in the gather (each a 16-byte slice of the input row)
  constexpr int NPW = 64 / LPN;       // nodes per wavefront and pass
  constexpr int NPP = 4 * NPW;        // nodes per workgroup pass
  constexpr int PASSES = 32 / NPP;
  constexpr int TRIP = FIN == 32 ? 4 : 8;   // in-edges per trip (slot q of a trip is evaluated by lane q of the node's group): the patch
                                      // grid has <= 4 per node (one trip, 16 row registers: three workgroups per CU); the 64-wide
                                      // layers (configs[3]: in-degree 8) take 8
  constexpr int P2 = 2 * H;
  constexpr int FO = 32 * NT;         // output width of a head
  constexpr int KS = FIN / 16;        // k steps of the bf16 MFMA
  // GEMM work split: wave w owns n tile w % NT of the heads g, g + NG, ... (g = w / NT): the heads of a wave are summed in REGISTERS
  // for the head mean, and only the NGE head groups meet in LDS (H = 4, F' = 64: two heads per wave, a two-way exchange)
  constexpr int NG = 4 / NT;          // head groups
  constexpr int HPW = (H + NG - 1) / NG;   // heads per wave
  constexpr int NGE = H < NG ? H : NG;     // groups that hold a head
  static_assert(H == 1 || H == 2 || H == 4, "heads");
  static_assert(NT == 1 || NT == 2, "n tiles");
  typedef float fH __attribute__((ext_vector_type(4)));   // H <= 4 head values of a node / edge (unused tail lanes stay 0)
  // aggregated input rows, already split into the three exact bf16 pieces of the GEMM's A operand (x3.h): [head][piece][node][FIN
  // bf16 + 16 bytes of pad] -- the split is done ONCE by the lane that produced the values (both n-tile waves of a head read them)
  constexpr int APITCH = FIN * 2 + 16;   // bytes per node row: 16-byte reads one node apart fall on distinct banks
  __shared__ __attribute__((aligned(16))) unsigned char agg_s[H][3][32 * APITCH];
  constexpr int RLD = 32 + 4;         // pitch of the head-mean exchange rows (floats)
  __shared__ __attribute__((aligned(16))) float exch_s[NT][NGE][32 * RLD];   // head-mean exchange: [n tile][head group][node][channel]
  __shared__ __attribute__((aligned(16))) float wts_s[32][TRIP][4];    // softmax weights of a trip: [node][slot][head]
  __shared__ __attribute__((aligned(16))) int cols_s[32][TRIP];        // source ids of a trip
  // the H attention scalars of a node (a row of st is 2H floats: 8-, 16- or 32-byte aligned) in one load
  auto load_heads = [](const float* p) {
    fH v = {0.f, 0.f, 0.f, 0.f};
    if constexpr (H == 4) {
      v = *reinterpret_cast<const fH*>(p);
    } else if constexpr (H == 2) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 t = *reinterpret_cast<const f32x2*>(p);
      v[0] = t[0], v[1] = t[1];
    } else {
      v[0] = p[0];
    }
    return v;
  };
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane % LPN, ln = lane / LPN;
  const int lr = lane & 31, lh = lane >> 5;
  const int wnt = wave % NT, wg = wave / NT;   // this wave's n tile and head group
  const bool gemm_wave = wg < H;               // the group holds at least one head

  // the weight pieces of this wave's (head, n tile) items (registers, loaded once per workgroup)
  u32x4 bw[HPW][KS][3];
#pragma unroll
  for (int k = 0; k < HPW; ++k)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const int item = min(wg + k * NG, H - 1) * NT + wnt;
        bw[k][ks][pc] = *reinterpret_cast<const u32x4*>(Wx + ((size_t)((item * KS + ks) * 3 + pc) * 64 + lane) * 4);
      }

  // XCD-aware persistent walk: workgroup b runs on XCD b % 8; each XCD owns a contiguous range of node tiles and its workgroups
  // take them round-robin, so the workgroups that share an L2 work on neighbouring patch rows (sources j +- 1, j +- npw)
  const int nwg = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, wgx = nwg >> 3;              // the launcher makes nwg a multiple of 8
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int t_count = tq + (xcd < tr ? 1 : 0);
  // graph of the current tile: tiles are walked in ascending order, so the id only moves forward (wave-uniform scalar loads of
  // graph_ptr -- no per-tile node -> graph load in front of everything else)
  int g_cur = 0, g_hi = (gp && G > 1) ? 0 : N;        // nodes below g_hi belong to graph g_cur (or an earlier one)
  int g_cached = -1;
  fH gm_cached = {0.f, 0.f, 0.f, 0.f};
  // Software pipeline over the tiles of this workgroup.  A tile's first touch of its CSR rows, attention scalars and source rows
  // is a chain of dependent misses (rowptr -> col -> {source row, s}: ~2000 cycles each under load, stamps of the diagnostic
  // build), and with three workgroups per CU nothing else covers it.  So everything a tile reads from global memory is requested
  // while EARLIER tiles compute:
  //   rows (rowptr, degree, t) of tile i + 2   at the top of tile i                 (meta_rows)
  //   first-trip source ids    of tile i + 2   behind the epilogue of tile i       (meta_cols)
  //   first-trip source rows + s of tile i + 1 behind the aggregate of tile i      (fetch_trip0: the row registers of tile i are
  //                                                                                 dead by then, so this costs no registers)
  // In the steady state a tile starts with its operands in registers; later trips of nodes with more than TRIP in-edges load in place.
  constexpr bool XPF = FIN == 32;   // source-row prefetch (the 64-wide layers would hold 2 passes x 8 rows = 64 more registers)
  struct Meta {
    int rs[PASSES], deg[PASSES], c0[PASSES];
    fH tj[PASSES];
  };
  Meta m_cur, m_nxt, m_nn;           // tile i, tile i + 1 (its c0 trails by half a tile), rows of tile i + 2
  auto meta_rows = [&](const int n0t, Meta& m, const bool with_t) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int node = n0t + p * NPP + wave * NPW + ln;
      const bool live = node < N;
      const int nd = live ? node : min(n0t, N - 1);
      m.rs[p] = rowptr[nd];
      // unconditional (nd + 1 <= N) and used unconditionally (a mask, not a select of the load: hipcc sinks a load whose value is
      // needed on one side only under the condition, and the branch carries a vmcnt(0) that drained the row prefetch issued before it)
      m.deg[p] = (rowptr[nd + 1] - m.rs[p]) & (live ? -1 : 0);
      if (with_t) m.tj[p] = load_heads(st + (size_t)nd * P2 + H);
    }
  };
  auto meta_t = [&](const int n0t, Meta& m) {   // the target-side attention scalars t: needed one tile ahead only
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int node = n0t + p * NPP + wave * NPW + ln;
      m.tj[p] = load_heads(st + (size_t)(node < N ? node : min(n0t, N - 1)) * P2 + H);
    }
  };
  auto meta_cols = [&](Meta& m) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) m.c0[p] = col[(q < TRIP && q < m.deg[p]) ? m.rs[p] + q : 0];
  };
  f32x4 xj0[XPF ? PASSES : 1][TRIP];   // first-trip source rows of the tile about to be aggregated
  fH sj0[XPF ? PASSES : 1];
  // the group's source ids through LDS, then every row load of the trip in flight (a missing edge re-reads col[0]'s row, weight 0)
  auto fetch_rows = [&](const int nl, const int c_own, f32x4 (&xr)[TRIP], fH& sr) {
    if (q < TRIP) cols_s[nl][q] = c_own;
    sr = load_heads(st + (size_t)c_own * P2);
    __builtin_amdgcn_wave_barrier();   // the group's lanes are in this wave and its LDS operations complete in order
    int cs[TRIP];
#pragma unroll
    for (int u4 = 0; u4 < TRIP; u4 += 4) {
      const int4 ca = *reinterpret_cast<const int4*>(&cols_s[nl][u4]);
      cs[u4] = ca.x, cs[u4 + 1] = ca.y, cs[u4 + 2] = ca.z, cs[u4 + 3] = ca.w;
    }
#pragma unroll
    for (int u = 0; u < TRIP; ++u)
      xr[u] = *reinterpret_cast<const f32x4*>(x + (size_t)cs[u] * FIN + 4 * q);
    __builtin_amdgcn_wave_barrier();   // the ids have been read: the next use of the group's line may overwrite them
  };
  auto fetch_trip0 = [&](const Meta& m) {
    if constexpr (XPF) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) fetch_rows(p * NPP + wave * NPW + ln, m.c0[p], xj0[p], sj0[p]);
    }
  };
  const int ti0 = b >> 3;
  auto tile_n0 = [&](const int ti) { return (t_begin + min(ti, t_count - 1)) * 32; };   // (past the end: the last tile again, never used)
  if (ti0 < t_count) {
    meta_rows(tile_n0(ti0), m_cur, true);
    meta_cols(m_cur);
    meta_rows(tile_n0(ti0 + wgx), m_nxt, true);
    fetch_trip0(m_cur);
    meta_cols(m_nxt);
  }
  m_nn = m_nxt;
  [[maybe_unused]] int dti = -1;
#if defined(MGU_DIAG) && MGU_DIAG == 41
  __shared__ unsigned long long diag_ts[4][8][16];
#endif
  for (int ti = ti0; ti < t_count; ti += wgx) {
    const int n0 = tile_n0(ti);
    ++dti;
    GAT_T(0);
    // rows of the tile after next: requested FIRST, a whole tile before their first use (the source ids read through them behind
    // this tile's GEMM): hipcc hoists the dependent address arithmetic as far up as it can, and with the request just in front of
    // barrier A the second MFMA of the GEMM waited for it -- and, vmcnt retiring in order, for the row prefetch in front of it
    meta_rows(tile_n0(ti + 2 * wgx), m_nn, false);
    if (n0 >= g_hi) {   // (first tile / a new graph) wave-uniform
      g_cur = gf_graph_of(gp, G, n0);
      g_hi = gp[g_cur + 1];
    }
    const bool one_graph = min(n0 + 31, N - 1) < g_hi;   // block-uniform
    // per-graph maxima of the H heads: re-read only when the tile's graph changes (one cooperative 64-slot read per head)
    if (one_graph && g_cur != g_cached) {
#pragma unroll
      for (int h = 0; h < H; ++h) gm_cached[h] = gmax_read_wave(gmax, gstride, g_cur * H + h, gen);
      g_cached = g_cur;
    }
    // ---- gather + softmax-weighted aggregate of the INPUT rows, all heads --------------------------------------------------
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int nl = p * NPP + wave * NPW + ln;        // node slot inside the tile
      const int rs = m_cur.rs[p], deg = m_cur.deg[p];
      const fH tj = m_cur.tj[p];
      fH gm = gm_cached;
      GAT_T(1);   // (gmax)
      if (!one_graph) {   // a tile that straddles graphs (rare on image grids; the rule for tiny graphs): per-lane entries
        const int g = gf_graph_of(gp, G, min(n0 + nl, N - 1));
#pragma unroll
        for (int h = 0; h < H; ++h) gm[h] = gmax_read_lane(gmax, gstride, g * H + h, gen);
      }
      f32x4 acc[H];
      fH D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < H; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int maxdeg = (int)wave_max_u32((unsigned)deg);
      const bool single = maxdeg <= TRIP;   // wave-uniform: every node of the wave is done in one trip (the patch grids)
      for (int e0 = 0; e0 < maxdeg; e0 += TRIP) {
        // slot q of the trip is evaluated by lane q of the node's group
        const bool own = q < TRIP && e0 + q < deg;
        f32x4 xj[TRIP];
        fH sj;
        if (XPF && e0 == 0) {
#pragma unroll
          for (int u = 0; u < TRIP; ++u) xj[u] = xj0[XPF ? p : 0][u];
          sj = sj0[XPF ? p : 0];
        } else {
          const int c_own = e0 == 0 ? m_cur.c0[p] : col[own ? rs + e0 + q : 0];
          fetch_rows(nl, c_own, xj, sj);
        }
        GAT_T(3);   // source rows
        fH w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float ev = sj[h] + tj[h];
          ev = fmaxf(ev, alpha * ev);                             // LeakyReLU (graph_attention.py:65), 0 <= alpha <= 1
          w[h] = own ? __expf(ev - gm[h]) : 0.f;                  // exp(e - max(e)) (:86): ONE evaluation per (edge, head)
        }
        if (single) {
          // all in-edges of a node sit in the TRIP owner lanes of its group: the segment sum D is a DPP fold over them and the
          // owners hand out the NORMALISED coefficients alpha = exp / (D + 1e-10) (:96), so the other lanes neither sum D nor divide
#pragma unroll
          for (int h = 0; h < H; ++h) {
            float d = w[h];
            d = dpp_add<0xB1>(d);                                 // quad_perm [1,0,3,2]
            d = dpp_add<0x4E>(d);                                 // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
            if (TRIP == 8) d = dpp_add<0x141>(d);                 // row_half_mirror: the two owner quads of a 16-lane group meet
            w[h] *= __builtin_amdgcn_rcpf(d + 1e-10f);
          }
        }
        if (q < TRIP) *reinterpret_cast<fH*>(&wts_s[nl][q][0]) = w;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
          const fH wu = *reinterpret_cast<const fH*>(&wts_s[nl][u][0]);
          if (!single) D += wu;
#pragma unroll
          for (int h = 0; h < H; ++h) acc[h] += wu[h] * xj[u];
        }
        __builtin_amdgcn_wave_barrier();   // the next trip rewrites the group's lines
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        f32x4 a = acc[h];
        if (!single) a = a * __builtin_amdgcn_rcpf(D[h] + 1e-10f);   // (:96); a node without in-edges aggregates to 0
        // three exact bf16 pieces of the lane's four values: 8 bytes per piece at [node][4 q .. 4 q + 3]
        unsigned p0a, p1a, p2a, p0b, p1b, p2b;
        split3_pack(a[0], a[1], p0a, p1a, p2a);
        split3_pack(a[2], a[3], p0b, p1b, p2b);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(&agg_s[h][0][nl * APITCH + 8 * q]) = u32x2{p0a, p0b};
        *reinterpret_cast<u32x2*>(&agg_s[h][1][nl * APITCH + 8 * q]) = u32x2{p1a, p1b};
        *reinterpret_cast<u32x2*>(&agg_s[h][2][nl * APITCH + 8 * q]) = u32x2{p2a, p2b};
      }
    }
    // the next tile's first-trip rows and s (its source ids arrived during the previous tile), then the rows of the tile after it
    m_cur = m_nxt;
    fetch_trip0(m_cur);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) m_nxt.rs[p] = m_nn.rs[p], m_nxt.deg[p] = m_nn.deg[p];
    meta_t(tile_n0(ti + 2 * wgx), m_nxt);   // (its t one tile ahead)
    GAT_T(4);   // weights, aggregate written
    lds_barrier();   // (A) the aggregate tile is complete; every wave has left the previous tile's exchange reads
    GAT_T(5);
    // ---- (32 nodes x FIN) . W_h^T (FIN x 32) per (head, n tile) item on the bf16 matrix cores, exact three-way splits --------
    f32x16 c[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const int hh = min(wg + k * NG, H - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 pa[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) pa[pc] = *reinterpret_cast<const u32x4*>(&agg_s[hh][pc][lr * APITCH + ks * 32 + lh * 16]);
        f32x16 t = c[k];
        t = mfma_bf16(pa[2], bw[k][ks][0], t);
        t = mfma_bf16(pa[0], bw[k][ks][2], t);
        t = mfma_bf16(pa[1], bw[k][ks][1], t);
        t = mfma_bf16(pa[1], bw[k][ks][0], t);
        t = mfma_bf16(pa[0], bw[k][ks][1], t);
        t = mfma_bf16(pa[0], bw[k][ks][0], t);
        c[k] = t;
      }
    }
    // first-trip source ids of the tile after next (its row pointers were requested before barrier A) -- BEFORE the output stores:
    // vmcnt retires in order and counts stores, so a load behind them waits for their HBM round trip when it is needed
    __builtin_amdgcn_sched_barrier(0);
    meta_cols(m_nxt);
    asm volatile("" ::"v"(c[0][0]));
    GAT_T(6);   // GEMM
    auto elu = [](const float v) { return v > 0.f ? v : (__expf(v) - 1.f); };   // (:118)
    // (inside a wave that holds a head, every one of its HPW slots holds one: H is a multiple of NG or smaller than it -- so the
    // epilogue below has no per-register conditions: with a test per value hipcc emitted a branch per value, and the 32 exp chains of
    // a lane ran one after the other, 2 500 cycles per tile)
    static_assert(H % NG == 0 || H < NG, "head groups are full");
    const bool full = n0 + 32 <= N;   // block-uniform: no per-row bound checks on a whole tile
    // the epilogue's per-lane constants from an OPAQUE copy of the thread id, so that none of its addresses is loop-invariant: hipcc
    // hoists them out of the tile loop otherwise and spills them around it (27 registers, one reload in front of the first MFMA)
    int et = threadIdx.x;
    asm volatile("" : "+v"(et));
    const int elr = et & 31, elh = (et >> 5) & 1;
    if (concat) {
      if (gemm_wave) {
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
          const int hh = wg + k * NG;
          float* const o = out + (size_t)n0 * (H * FO) + hh * FO + wnt * 32 + elr;   // cat over heads (:155)
#pragma unroll
          for (int r4 = 0; r4 < 16; r4 += 4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = elu(c[k][r4 + e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = e + 8 * (r4 >> 2) + 4 * elh;
              if (full || n0 + row < N) o[(size_t)row * (H * FO)] = v[e];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      lds_barrier();   // (B) every wave has read its A operands: the next tile's aggregate may overwrite them
    } else {
      // mean over heads (:158): the heads of this wave are added in registers (fixed order); the NGE head groups meet in LDS as rows
      // [n tile][group][node][32 channels + pad], and thread (node, channel quad) adds the groups in the order 0 .. NGE - 1 and
      // stores 16 bytes per n tile (a row of the output is written as whole 128-byte lines by eight threads): bitwise reproducible
      if (gemm_wave) {
        // four registers at a time (the exp chains of a group interleave; all 32 at once cost 27 spilled registers)
#pragma unroll
        for (int r4 = 0; r4 < 16; r4 += 4) {
#pragma unroll
          for (int r = r4; r < r4 + 4; ++r) {
            float sum = elu(c[0][r]);
#pragma unroll
            for (int k = 1; k < HPW; ++k) sum += elu(c[k][r]);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * elh;
            exch_s[wnt][wg][row * RLD + elr] = sum;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      lds_barrier();   // (B)
      constexpr float invH = 1.f / H;
      const int row = et >> 3, qd = et & 7;        // 32 rows x 8 channel quads = 256 threads
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 sum = *reinterpret_cast<const f32x4*>(&exch_s[nt][0][row * RLD + 4 * qd]);
#pragma unroll
        for (int g2 = 1; g2 < NGE; ++g2) sum += *reinterpret_cast<const f32x4*>(&exch_s[nt][g2][row * RLD + 4 * qd]);
        if (full || n0 + row < N)
          *reinterpret_cast<f32x4*>(out + (size_t)(n0 + row) * FO + nt * 32 + 4 * qd) = sum * invH;
      }
    }
    GAT_T(7);   // ELU, head mean, stores
  }
#if defined(MGU_DIAG) && MGU_DIAG == 41
  if (blockIdx.x == 100 && lane == 0)
    for (int t = 0; t <= min(dti, 7); ++t)
      for (int k = 0; k < 16; ++k) mgu_diag_gat[wave][t][k] = diag_ts[wave][t][k];
#endif
}
#if defined(MGU_DIAG) && (MGU_DIAG == 40 || MGU_DIAG == 41)
}  // namespace mgu
extern "C" int mgu_diag_gat_read(unsigned long long* out, int n) {
  if (n > 4 * 8 * 16) n = 4 * 8 * 16;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgu::mgu_diag_gat), (size_t)n * sizeof(unsigned long long));
}
namespace mgu {
#endif

bool gat_fused_applicable(int Fin, int heads, int Fh, int64_t E) {
  return (Fin == 32 || Fin == 64) && (Fh == 32 || Fh == 64) && (heads == 1 || heads == 2 || heads == 4) && Fin <= Fh && E > 0;
}

template <int FIN, int NT, int H>
static hipError_t launch_fused2_t(const float* x, const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* gp, int G,
                                  const unsigned long long* gmax, const unsigned* Wx, int N, int concat, float alpha, float* out, int gstride,
                                  unsigned gen, hipStream_t s) {
  const int ntiles = (N + 31) / 32;
  // persistent workgroups: as many as are resident at once (occupancy x CUs), a multiple of 8 so that every XCD owns the same
  // number; small batches get one tile per workgroup
  // resident workgroups of THIS kernel instance on THIS device: a per-device table of values that are a pure function of (instance,
  // device) -- filled on first use, identical whichever thread fills it (relaxed atomics: no ordering is needed, only no torn value)
  static std::atomic<int> resident_of[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& slot = resident_of[dev >= 0 && dev < 64 ? dev : 0];
  int resident = slot.load(std::memory_order_relaxed);
  if (!resident) {
    int cus = 256, occ = 2;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&gat_fused2_kernel<FIN, NT, H>), 256, 0) != hipSuccess || occ < 1)
      occ = 2;
    resident = std::max(8, cus * occ / 8 * 8);
    slot.store(resident, std::memory_order_relaxed);
  }
  const int nwg = std::min((ntiles + 7) / 8 * 8, resident);
  hipLaunchKernelGGL((gat_fused2_kernel<FIN, NT, H>), dim3(nwg), dim3(256), 0, s, x, st, rowptr, col, gp, G, gmax, Wx, N, ntiles,
                     concat, alpha, out, gstride, gen);
  return hipGetLastError();
}

hipError_t launch_gat_fused(const float* x, int Fin, const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* gp, int G,
                            const unsigned long long* gmax, const unsigned* Wx, int N, int heads, int Fh, int concat, float alpha, float* out,
                            int gstride, unsigned gen, hipStream_t s) {
  if (N == 0) return hipSuccess;
#define MGU_GF2(FIN, NT, H) return launch_fused2_t<FIN, NT, H>(x, st, rowptr, col, gp, G, gmax, Wx, N, concat, alpha, out, gstride, gen, s)
#define MGU_GF2_H(FIN, NT)               \
  do {                                   \
    if (heads == 1) MGU_GF2(FIN, NT, 1); \
    if (heads == 2) MGU_GF2(FIN, NT, 2); \
    if (heads == 4) MGU_GF2(FIN, NT, 4); \
  } while (0)
  if (Fin == 32 && Fh == 32) MGU_GF2_H(32, 1);
  if (Fin == 32 && Fh == 64) MGU_GF2_H(32, 2);
  if (Fin == 64 && Fh == 64) MGU_GF2_H(64, 2);
#undef MGU_GF2_H
#undef MGU_GF2
  return hipErrorInvalidValue;
}

}  // namespace mgu
