// Graph attention message passing for gfx950: replaces GraphAttentionLayer.forward
// (model/gat/graph_attention.py:40-118) and the per-head Python loop of MultiHeadGATLayer (:150-160).
//
// The reference gathers 2*E rows of Wh to build e = LeakyReLU(a . [Wh_src || Wh_tgt]) (:57-65).  That
// dot product splits into per-NODE scalars s_i = a[:F'] . Wh_i and t_j = a[F':] . Wh_j, so an edge only
// needs two scalars for its logit and ONE row gather (Wh_src) for the aggregate.  All heads are done in
// one pass: a row of Wh is heads*F' floats (1 KiB for 4x64), i.e. exactly one 16-byte-per-lane
// coalesced wavefront load.  The softmax is the reference's literal form: exp(e - GLOBAL max over the
// graph's edges) (:86), divided by (segment sum + 1e-10) (:96) -- not a per-target stabilised softmax.
//
//   (igemm)       : Whp = X [W | W^T a_src | W^T a_tgt]^T : node table rows [Wh (H*F') | s (H) | t (H)]
//   gat_edge_max  : per (graph, head) max_e                        (CSR + s,t only; order-encoded atomicMax)
//   gat_aggregate : CSR-by-target neighbour gather -> weighted sum -> /(D+1e-10) -> ELU -> concat | head-mean
//                   one wavefront per 4-row CSR segment, 8 row gathers in flight per lane, segmented
//                   register accumulation, head-mean staged through LDS, XCD-aware workgroup order.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gat_common.h"

namespace mgu {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned enc_ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ int graph_of(const int32_t* __restrict__ gp, int G, int node) {
  if (!gp || G <= 1) return 0;
  int lo = 0, hi = G;  // gp[lo] <= node < gp[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gp[mid] <= node) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- node -> graph id table (one coalesced kernel instead of a dependent binary search per wavefront) ----
__global__ void gat_node_graph_kernel(const int32_t* __restrict__ gp, int G, int nodes_per_graph, int N,
                                      int32_t* __restrict__ node_graph) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < N) node_graph[j] = nodes_per_graph > 0 ? min(j / nodes_per_graph, G - 1) : graph_of(gp, G, j);
}

hipError_t launch_gat_node_graph(const int32_t* gp, int G, int nodes_per_graph, int N, int32_t* node_graph, hipStream_t s) {
  hipLaunchKernelGGL(gat_node_graph_kernel, dim3((N + 255) / 256), dim3(256), 0, s, gp, G, nodes_per_graph, N, node_graph);
  return hipGetLastError();
}

// ---- extra panel rows so the linear GEMM also emits s and t -------------------------------------------
// s_i[h] = a[h][:Fh] . (W_h x_i) = (W_h^T a[h][:Fh]) . x_i: rows HF+h (s) and HF+H+h (t) of the weight panel
// hold W_h^T a_src / W_h^T a_tgt, so ONE GEMM emits the node table Wh (N, HF) and, through the split-column
// epilogue, the compact attention-scalar table st (N, 2H) = [s | t] (graph_attention.py:53,57-64).
constexpr int GAT_MAX_HF = 1024;

__global__ __launch_bounds__(256) void gat_wa_rows_kernel(const float* __restrict__ W, const float* __restrict__ a,
                                                          float* __restrict__ panel, int row0, int heads, int Fh, int Fin, int Kp) {
  // one workgroup per output row r = (s|t, head); thread (fl, k): partial dot over f = fl, fl+nfl, ...; LDS fold
  __shared__ float red[256];
  const int r = blockIdx.x;
  const int which = r / heads, h = r - which * heads;
  const int kw = min(Fin, 256), nfl = 256 / kw;
  const int t = threadIdx.x, kl = t % kw, fl = t / kw;
  for (int k0 = 0; k0 < Fin; k0 += kw) {
    const int k = k0 + kl;
    float sum = 0.f;
    if (fl < nfl && k < Fin)
      for (int f = fl; f < Fh; f += nfl) sum += a[h * 2 * Fh + which * Fh + f] * W[(size_t)(h * Fh + f) * Fin + k];
    red[t] = sum;
    __syncthreads();
    if (fl == 0 && k < Fin) {
      float tot = 0.f;
      for (int i = 0; i < nfl; ++i) tot += red[i * kw + kl];
      panel[(size_t)(row0 + r) * Kp + k] = tot;
    }
    __syncthreads();
  }
}

// rows row0 .. row0 + 2*heads of `panel` (pitch Kp): row0 = heads*Fh appends them to the linear layer's weight panel,
// row0 = 0 with Kp = Fin gives the compact (2H, Fin) matrix of the aggregate-first path (gat_fused.hip)
hipError_t launch_gat_wa_rows(const float* W, const float* a, float* panel, int row0, int heads, int Fh, int Fin, int Kp, hipStream_t s) {
  hipLaunchKernelGGL(gat_wa_rows_kernel, dim3(2 * heads), dim3(256), 0, s, W, a, panel, row0, heads, Fh, Fin, Kp);
  return hipGetLastError();
}

// ---- per (graph, head): max over edges of LeakyReLU(s_src + t_tgt)  (torch.max(e), :86) -----------
__global__ __launch_bounds__(256) void gat_edge_max_kernel(const float* __restrict__ st,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ node_graph, int N, int heads,
                                                           float alpha, gmax_t* __restrict__ gmax, int gstride, unsigned gen) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave_first = j - lane;
  if (wave_first >= N) return;  // wave-uniform
  const bool valid = j < N;
  const int g = node_graph ? node_graph[valid ? j : wave_first] : 0;
  const int g0 = __builtin_amdgcn_readfirstlane(g);
  const bool uniform = __all(g == g0);
  const int P = 2 * heads;
  const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & (GMAX_SLOTS - 1);   // this wave's accumulator line (gat_common.h)
  const float* s = st;
  const float* t = st + heads;
  int start = 0, end = 0;
  if (valid) {
    start = rowptr[j];
    end = rowptr[j + 1];
  }
  for (int h = 0; h < heads; ++h) {
    float m = -INFINITY;
    for (int k = start; k < end; ++k) m = fmaxf(m, s[(size_t)col[k] * P + h]);
    float e = -INFINITY;
    if (end > start) {
      e = m + t[(size_t)j * P + h];
      e = e > 0.f ? e : alpha * e;  // LeakyReLU is monotone: max commutes with it
    }
    if (uniform) {
      e = wave_max_f32(e);
      if (lane == 0 && e > -INFINITY) gmax_add(gmax, gstride, slot, g0 * heads + h, gen, e);
    } else if (e > -INFINITY) {
      gmax_add(gmax, gstride, slot, g * heads + h, gen, e);
    }
  }
}

hipError_t launch_gat_edge_max(const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* node_graph, int N,
                               int heads, float alpha, unsigned long long* gmax_enc, int gstride, unsigned gen, hipStream_t s) {
  if (N == 0) return hipSuccess;
  hipLaunchKernelGGL(gat_edge_max_kernel, dim3((N + 255) / 256), dim3(256), 0, s, st, rowptr, col, node_graph, N, heads, alpha,
                     gmax_enc, gstride, gen);
  return hipGetLastError();
}

// ---- neighbour gather + attention-weighted aggregate + normalise + ELU + concat/mean --------------
// One wavefront owns R consecutive target rows (a CSR segment of R rows).  The neighbour ids of the whole
// segment are fetched with one coalesced load (one id per lane); source rows are then gathered EIGHT AT A
// TIME (8 independent 16-byte-per-lane loads in flight per lane = 8 KiB per wave) before any of them is
// consumed, and each edge is accumulated into the register accumulator of the row that owns it (segmented
// reduction by a wave-uniform row index).  The head mean is staged through a per-wave LDS row.  Workgroup
// ids are remapped so each XCD (private 4 MiB L2) works on a contiguous range of rows: neighbouring patch
// rows (j +- 1, j +- npw) then hit the same L2 instead of being re-fetched by other XCDs.
// DENSE: also compile the 2 x 8 batch of the lane-parallel fast path (graphs with more than 4 in-edges per node on average); the
// sparse instantiation stays at 92 registers = 5 waves per SIMD, which the patch grids need (26.4 -> 23.6 us at 64 graphs)
template <int NCH, bool DENSE = false>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(const float* __restrict__ wh, int P, const float* __restrict__ st,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ node_graph,
                                                            const gmax_t* __restrict__ gmax, int N, int heads, int Fh,
                                                            int concat, float alpha, float* __restrict__ out, int gstride, unsigned gen) {
  constexpr int R = 4;    // rows (one CSR segment) per wavefront (8 rows = two 4 x 4 batches per wave measured no faster: 24.7 vs 23.6 us)
  constexpr int EB = 8;   // row gathers in flight per lane
  __shared__ __attribute__((aligned(16))) float stage[4][NCH == 1 ? 4 * 256 + 64 : NCH * 256];   // NCH 1: a wave's four rows + its 64 softmax weights
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware bijective remap of the workgroup id (blocks b, b+8, ... share an XCD)
  const int nblk = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, qn = nblk >> 3, rn = nblk & 7;
  const int blk = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
  const int j0 = (blk * 4 + wave) * R;
  if (j0 >= N) return;  // wave-uniform; no block-level barrier below
  const int HF = heads * Fh, nq = HF >> 2, qh = Fh >> 2, H2 = 2 * heads;
  const float inv_heads = 1.f / (float)heads;

  // CSR segment of the R rows: lanes 0..R hold rowptr[j0 .. j0+R]; each row's neighbour ids sit one per lane
  const int rpv = rowptr[min(j0 + min(lane, R), N)];
  const int start = __builtin_amdgcn_readlane(rpv, 0);
  const int ne = __builtin_amdgcn_readlane(rpv, R) - start;
  const int seg_ids = (lane < ne) ? col[start + lane] : 0;   // whole segment when it has <= 64 edges (the usual case)

  int head[NCH];
  unsigned coff4[NCH];
  bool on[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    on[i] = c < nq;
    head[i] = on[i] ? c / qh : 0;
    coff4[i] = on[i] ? (unsigned)c * 4u : 0u;
  }

  // per-(graph, head) max: when the segment's rows lie in ONE graph (wave-uniform, the usual case) each head's 64 slots are read
  // once by the whole wave; a lane then keeps the value of the head its channels belong to
  const int gs0 = node_graph ? node_graph[j0] : 0, gs1 = node_graph ? node_graph[min(j0 + R - 1, N - 1)] : 0;
  const bool seg_one_graph = gs0 == gs1;

  // ---- fast path, lane-parallel attention (every patch-graph segment): all R rows have <= 4 in-edges, lie in one graph, and
  // the layer has <= 4 heads -----------------------------------------------------------------------------------------------------
  // The kernel is instruction-issue bound (rocprofv3, 64 graphs: 653 VALU + 364 SALU per 4-row segment, 34 % of the wave cycles
  // issuing, and the same 0.49 ns per node whether the node table sits in L2, in the Infinity Cache or in HBM).  Most of those
  // instructions were REDUNDANT: the softmax weight x = exp(LeakyReLU(s + t) - max) of an (edge, head) pair was evaluated by all
  // 16 lanes that hold the head's channels, for each of the 16 (row, edge slot) pairs -- 16 x 16 evaluations per lane where the
  // segment has 64 distinct values.  Here lane (head h = lane / 16, pair p = lane % 16 = row * 4 + slot) evaluates ITS value
  // once (one scalar gather of s, one exp), the 64 weights go through a 256-byte LDS line, and each channel lane reads the 16
  // weights of its head back with four 16-byte reads.  The per-graph max is reduced inside the 16-lane row of a head (four DPP
  // steps over 4 x 16 slots) instead of four full-wave reductions.
  if constexpr (NCH == 1)
  if (heads <= 4 && seg_one_graph && j0 + R <= N && ne <= 64) {
    int s0r[R], dg[R], dmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s0r[r] = __builtin_amdgcn_readlane(rpv, r) - start;
      dg[r] = __builtin_amdgcn_readlane(rpv, r + 1) - start - s0r[r];
      dmax = max(dmax, dg[r]);
    }
    // a batch = RB rows x K edge slots = 16 (row, slot) pairs per head: 4 x 4 (the patch grids) or 2 x 8 (denser graphs, e.g. the
    // in-degree-8 stress graphs of BASELINE configs[3]), two batches per segment in the latter case
    auto fast = [&](auto rb_c, auto k_c) {
      constexpr int RB = decltype(rb_c)::value, K = decltype(k_c)::value;
      static_assert(RB * K == 16, "16 pairs per head");
      float* const xs = &stage[wave][4 * 256];        // 64 weights behind this wave's (<= 4) head-mean rows
      // max of head hh over its 64 slots: 4 slots per lane of the head's 16-lane row, then a row reduction (once per segment)
      const int ph = lane >> 4, pp = lane & 15, pr = pp / K, pk = pp % K;
      const int hh = min(ph, heads - 1);
      unsigned um = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const gmax_t w = gmax[(size_t)(gs0 * heads + hh) * GMAX_SLOTS + q * 16 + pp];
        um = max(um, (unsigned)(w >> 32) == gen ? (unsigned)w : 0u);
      }
      um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
      um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
      um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0x141, 0xF, 0xF, true));   // row_half_mirror
      um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0x140, 0xF, 0xF, true));   // row_mirror
      const float pgm = um ? gat_dec_ordered(um) : -INFINITY;
#pragma unroll
      for (int rb = 0; rb < R; rb += RB) {
        // -- producer: lane (h, p) evaluates the weight of pair p = (row rb + pr, slot pk) for head h --
        const int ra = rb + pr;
        int pdg = dg[rb], ps0 = s0r[rb];
#pragma unroll
        for (int r = 1; r < RB; ++r) pdg = pr == r ? dg[rb + r] : pdg, ps0 = pr == r ? s0r[rb + r] : ps0;
        // (the shuffle runs UNCONDITIONALLY: inside `pdg > 0 ? ... : j0` hipcc branches around it, and a ds_bpermute under a
        //  partial EXEC mask cannot read the lanes that are switched off -- exactly the lanes of an empty row, which hold ids)
        const int pany = __shfl(seg_ids, ps0 + min(pk, max(pdg, 1) - 1));
        const int psrc = pdg > 0 ? pany : j0;                                      // slots past the degree re-read the last neighbour
        const float psv = st[(unsigned)(psrc * H2) + hh];
        const float ptj = st[(size_t)(j0 + ra) * H2 + heads + hh];
        float pev = psv + ptj;
        pev = pev > 0.f ? pev : alpha * pev;                                        // LeakyReLU (:65)
        if (rb) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");             // the previous batch's reads of xs come first
        xs[lane] = (pk < pdg && ph < heads) ? __expf(pev - pgm) : 0.f;              // exp(e - max(e)) (:86); dead slots weigh 0
        // -- the 16 source-row gathers of the batch, all in flight before any is consumed --
        f32x4 v[RB][K];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int src = dg[rb + r] > 0 ? __builtin_amdgcn_readlane(seg_ids, s0r[rb + r] + min(k, dg[rb + r] - 1)) : j0;
            v[r][k] = *reinterpret_cast<const f32x4*>(wh + (unsigned)(src * P) + coff4[0]);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // a wavefront's LDS operations complete in order
        // -- consumer: a channel lane reads the 16 weights of its head --
        f32x4 xw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xw[q] = *reinterpret_cast<const f32x4*>(xs + head[0] * 16 + q * 4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          float D = 0.f;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float x = xw[(r * K + k) >> 2][(r * K + k) & 3];
            D += x;
            acc += x * v[r][k];
          }
          if (on[0]) {
            const float inv = __frcp_rn(D + 1e-10f);                          // (:96)
            f32x4 o = acc * inv;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : (__expf(o[q]) - 1.f);   // ELU (:118)
            if (concat) *reinterpret_cast<f32x4*>(out + (size_t)(j0 + rb + r) * HF + lane * 4) = o;
            else *reinterpret_cast<f32x4*>(&stage[wave][r * 256 + lane * 4]) = o;
          }
        }
        if (!concat) {   // head mean (:158): the batch's rows through the wave's own LDS stage
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          for (int u = lane; u < RB * qh; u += 64) {
            const int r = u / qh, c = u - r * qh;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            for (int h = 0; h < heads; ++h) sum += *reinterpret_cast<const f32x4*>(&stage[wave][r * 256 + h * Fh + c * 4]);
            *reinterpret_cast<f32x4*>(out + (size_t)(j0 + rb + r) * Fh + c * 4) = sum * inv_heads;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    };
    if (dmax <= 4) {   // wave-uniform
      fast(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
      return;
    }
    if constexpr (DENSE)
      if (dmax <= 8) {
        fast(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{});
        return;
      }
  }
  float gm_seg[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) gm_seg[i] = 0.f;
  if (seg_one_graph)
    for (int h = 0; h < heads; ++h) {
      const float mh = gmax_read_wave(gmax, gstride, gs0 * heads + h, gen);
#pragma unroll
      for (int i = 0; i < NCH; ++i) gm_seg[i] = head[i] == h ? mh : gm_seg[i];
    }

#pragma unroll 1
  for (int r = 0; r < R; ++r) {
    const int j = j0 + r;
    if (j >= N) break;                         // wave-uniform
    const int s0 = __builtin_amdgcn_readlane(rpv, r);
    const int deg = __builtin_amdgcn_readlane(rpv, r + 1) - s0;
    float tj[NCH], gm[NCH], D[NCH];
    f32x4 acc[NCH];
    const int g = node_graph ? node_graph[j] : 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      tj[i] = st[(size_t)j * H2 + heads + head[i]];
      gm[i] = seg_one_graph ? gm_seg[i] : gmax_read_lane(gmax, gstride, g * heads + head[i], gen);
      D[i] = 0.f;
      acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The kernel is instruction-issue bound (rocprofv3: ~350 VALU+SALU per row at 36 VGPRs), so a row of <= 4
    // in-edges (every patch-graph row) takes a 4-slot batch and only longer rows the 8-slot one: dead slots cost
    // full gathers and FMAs.
    auto batch = [&](auto eb_c, const int ids, const int lane0, const int eb, const int cnt) {
      constexpr int EBN = decltype(eb_c)::value;
      f32x4 v[EBN][NCH];
      float sv[EBN][NCH];
      // NO branch around the gathers (a conditional load makes hipcc wait per element): slots past the end of the
      // row re-read its last neighbour and get weight 0.
#pragma unroll
      for (int k = 0; k < EBN; ++k) {
        const int src = __builtin_amdgcn_readlane(ids, lane0 + min(eb + k, cnt - 1));
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          // 32-bit indices (the launcher guarantees N*P < 2^31): the row base is a scalar multiply
          v[k][i] = *reinterpret_cast<const f32x4*>(wh + (unsigned)(src * P) + coff4[i]);
          sv[k][i] = st[(unsigned)(src * H2) + head[i]];
        }
      }
#pragma unroll
      for (int k = 0; k < EBN; ++k) {
        const bool live = eb + k < cnt;        // wave-uniform
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          float ev = sv[k][i] + tj[i];
          ev = ev > 0.f ? ev : alpha * ev;                        // LeakyReLU (:65)
          const float x = (live && on[i]) ? __expf(ev - gm[i]) : 0.f;   // exp(e - max(e)) (:86)
          D[i] += x;                                              // scatter_add of exp_e (:90-91)
          acc[i] += x * v[k][i];                                  // scatter_add of alpha*Wh_src (:104-112)
        }
      }
    };
#pragma unroll 1
    for (int base = 0; base < deg; base += 64) {
      const bool in_seg = ne <= 64;            // wave-uniform: ids already in seg_ids at lane (s0 - start) + e
      const int ids = in_seg ? seg_ids : ((base + lane < deg) ? col[s0 + base + lane] : 0);
      const int lane0 = in_seg ? s0 - start : 0;
      const int cnt = min(64, deg - base);
      if (cnt <= 4) {
        batch(std::integral_constant<int, 4>{}, ids, lane0, 0, cnt);
      } else {
#pragma unroll 1
        for (int eb = 0; eb < cnt; eb += EB) batch(std::integral_constant<int, EB>{}, ids, lane0, eb, cnt);
      }
    }
    // /(D + 1e-10) (:96), ELU (:118), concat (:155) or head mean (:158)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (on[i]) {
        const float inv = __frcp_rn(D[i] + 1e-10f);
        f32x4 o = acc[i] * inv;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : (__expf(o[q]) - 1.f);
        if (concat)
          *reinterpret_cast<f32x4*>(out + (size_t)j * HF + (lane + 64 * i) * 4) = o;
        else
          *reinterpret_cast<f32x4*>(&stage[wave][(lane + 64 * i) * 4]) = o;
      }
    }
    if (!concat) {
      // only this wave touches stage[wave]; a wavefront's LDS operations complete in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      for (int c = lane; c < qh; c += 64) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int h = 0; h < heads; ++h) sum += *reinterpret_cast<const f32x4*>(&stage[wave][h * Fh + c * 4]);
        *reinterpret_cast<f32x4*>(out + (size_t)j * Fh + c * 4) = sum * inv_heads;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}


hipError_t launch_gat_aggregate(const float* wh, int P, const float* st, const int32_t* rowptr, const int32_t* col,
                                const int32_t* node_graph, const unsigned long long* gmax_enc, int N, int64_t E, int heads, int Fh,
                                int concat, float alpha, float* out, int gstride, unsigned gen, hipStream_t s) {
  const int HF = heads * Fh;
  if (HF > GAT_MAX_HF || (Fh & 3) || (P & 3) || (long)N * P >= (1l << 31)) return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  dim3 block(256);
#define MGU_AGG(NCH)                                                                                              \
  hipLaunchKernelGGL(gat_aggregate_kernel<NCH>, dim3((N + 15) / 16), block, 0, s, wh, P, st, rowptr, col, node_graph, gmax_enc, \
                     N, heads, Fh, concat, alpha, out, gstride, gen)
  if (HF <= 256 && E > 4 * (int64_t)N)
    hipLaunchKernelGGL((gat_aggregate_kernel<1, true>), dim3((N + 15) / 16), block, 0, s, wh, P, st, rowptr, col, node_graph, gmax_enc, N, heads,
                       Fh, concat, alpha, out, gstride, gen);
  else if (HF <= 256) MGU_AGG(1);
  else if (HF <= 512) MGU_AGG(2);
  else MGU_AGG(4);
#undef MGU_AGG
  return hipGetLastError();
}

}  // namespace mgu
