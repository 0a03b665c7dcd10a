// GAT layer schedule behind the C-ABI (include/mgunet.h): replaces GraphAttentionLayer / MultiHeadGATLayer.forward
// (model/gat/graph_attention.py:40-118, 150-160), eval mode.  Host orchestration only; kernels in gat_fused.hip, gat.hip, igemm.hip.
//
// Launches per layer call with prepared weights (mgu_gat_prepare, once per weight version):
//   Fin <= F' (the patch GAT 32 -> 4 x 64, the stress graph 64 -> 4 x 64): gat_stmax (st + per-graph max) -> gat_fused  = 2
//   otherwise (a concat hidden layer):  GEMM [Wh | s | t] -> gat_edge_max -> gat_aggregate                              = 3
// No memset nodes: the per-(graph, head) max accumulators carry a generation number (gat_common.h), so stale words lose every max.
#include <algorithm>

#include "ctx.h"

using namespace mgu;
using namespace mgud;

struct mgu_gat_weights {
  int heads = 0, Fh = 0, Fin = 0;
  bool fused = false;
  float* buf = nullptr;   // fused: [wa (2H, Fin) | Wx (three bf16 pieces of W^T, fragment order)];  gather: packed panel [NPp][Kp]
  int Kp = 0, NPp = 0;
};

namespace {

int prepare_into(mgu_ctx* c, mgu_gat_weights* p, const float* W, const float* a, int heads, int Fh, int Fin, int64_t E_hint, hipStream_t s) {
  const int HF = heads * Fh;
  p->heads = heads, p->Fh = Fh, p->Fin = Fin;
  p->fused = c->tn.gat_fused && gat_fused_applicable(Fin, heads, Fh, E_hint);
  size_t floats;
  if (p->fused) {
    floats = gat_fused_scratch_floats(Fin, heads, Fh);
  } else {
    p->Kp = rup(Fin, 32), p->NPp = rup(HF + 2 * heads, 128);
    floats = (size_t)p->NPp * p->Kp;
  }
  if (p->buf) (void)hipFree(p->buf);
  p->buf = nullptr;
  hipError_t e = hipMalloc((void**)&p->buf, floats * sizeof(float));
  if (e != hipSuccess) return fail(c, MGU_ERR_NOMEM, "hipMalloc(%zu) failed: %s", floats * sizeof(float), hipGetErrorString(e));
  if (p->fused) {
    HIPCHK(c, launch_gat_prep(W, a, p->buf, reinterpret_cast<unsigned*>(p->buf + (size_t)2 * heads * Fin), heads, Fh, Fin, s));
  } else {
    // nn.Linear weight (F',Fin) stacked over heads is already the [N][K] panel (K padded to 32); rows HF.. hold
    // W^T a_src / W^T a_tgt so the same GEMM emits the attention scalars s, t (graph_attention.py:53,57-64)
    HIPCHK(c, hipMemsetAsync(p->buf, 0, floats * sizeof(float), s));
    HIPCHK(c, launch_pack_conv_w(W, p->buf, 0, HF, Fin, Fin, 1, p->Kp, s));
    HIPCHK(c, launch_gat_wa_rows(W, a, p->buf, HF, heads, Fh, Fin, p->Kp, s));
  }
  return MGU_OK;
}

int check_layer_shape(mgu_ctx* c, int Fin, int heads, int Fh) {
  if (heads < 1 || heads > 32 || Fh < 4 || (Fh & 3) || Fin < 4 || (Fin & 3))
    return fail(c, MGU_ERR_INVALID, "GAT layer needs Fin %% 4 == 0, Fout_head %% 4 == 0, 1 <= heads <= 32 (Fin=%d Fout=%d heads=%d)", Fin,
                Fh, heads);
  if (heads * Fh > 1024) return fail(c, MGU_ERR_INVALID, "heads*Fout_head = %d exceeds 1024", heads * Fh);
  return MGU_OK;
}

}  // namespace

// the slotted per-(graph, head) max accumulators [64 slots][cap] of 64-bit (generation, value) words and this call's generation
int mgud::gmax_buffer(mgu_ctx* c, int need, unsigned long long** buf, unsigned* gen) {
  if (need > c->gmax_cap) {
    const int cap = (std::max(need, 256) + 15) / 16 * 16;   // entries per slot: whole 128-byte lines
    if (c->gmaxbuf) {
      HIPCHK(c, hipDeviceSynchronize());
      HIPCHK(c, hipFree(c->gmaxbuf));
      c->gmaxbuf = nullptr;
    }
    HIPCHK(c, hipMalloc((void**)&c->gmaxbuf, (size_t)64 * cap * sizeof(unsigned long long)));
    HIPCHK(c, hipMemset(c->gmaxbuf, 0, (size_t)64 * cap * sizeof(unsigned long long)));
    c->gmax_cap = cap, c->gmax_gen = 0;
  }
  if (c->gmax_gen == 0xffffffffu) {   // generation wrap: start over from a cleared array
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemset(c->gmaxbuf, 0, (size_t)64 * c->gmax_cap * sizeof(unsigned long long)));
    c->gmax_gen = 0;
  }
  *buf = c->gmaxbuf;
  *gen = ++c->gmax_gen;
  return MGU_OK;
}

// Wh (N, HF) and the attention scalars st (N, 2H) = [s | t] from ONE GEMM on the panel [W | W^T a_src | W^T a_tgt] (the gather
// schedule's first launch), with the panel packed on the spot into a buffer of the backward's own: mgu_gat_layer_backward
// recomputes the forward's intermediates from the CURRENT weights.
int mgud::gat_linear_st(mgu_ctx* c, const float* X, int N, int Fin, const float* W, const float* a, int heads, int Fh, float* wh, float* st,
                        hipStream_t s) {
  const int HF = heads * Fh, Kp = rup(Fin, 32), NPp = rup(HF + 2 * heads, 128);
  int rc = ensure(c, &c->gbpanel, &c->gbpanel_bytes, (size_t)NPp * Kp * sizeof(float));
  if (rc) return rc;
  float* panel = (float*)c->gbpanel;
  HIPCHK(c, hipMemsetAsync(panel, 0, (size_t)NPp * Kp * sizeof(float), s));
  HIPCHK(c, launch_pack_conv_w(W, panel, 0, HF, Fin, Fin, 1, Kp, s));
  HIPCHK(c, launch_gat_wa_rows(W, a, panel, HF, heads, Fh, Fin, Kp, s));
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = X, d.w = panel, d.out = wh;
  d.M = N, d.H = 1, d.W = N;
  d.Cp = Fin, d.ldin = Fin, d.KS = 1, d.K = Fin, d.Kp = Kp;
  d.N = HF + 2 * heads, d.ldout = HF;
  d.split_n = HF, d.out2 = st, d.ld2 = 2 * heads;
  HIPCHK(c, launch_igemm_f32(d, s));
  return MGU_OK;
}

namespace {

int forward_prepared(mgu_ctx* c, const mgu_gat_weights* p, const float* X, int N, const int32_t* rowptr, const int32_t* col, int64_t E,
                     const int32_t* graph_ptr, int num_graphs, int concat, float alpha, float* out, hipStream_t s) {
  const int heads = p->heads, Fh = p->Fh, Fin = p->Fin, HF = heads * Fh;
  if (num_graphs < 1 || !graph_ptr) num_graphs = 1, graph_ptr = nullptr;
  unsigned long long* gmax;
  unsigned gen;
  int rc = gmax_buffer(c, num_graphs * heads, &gmax, &gen);
  if (rc) return rc;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  if (p->fused && E > 0) {
    // aggregate-first path (gat_fused.hip): no (N, heads*F') node table, the gather moves Fin floats per edge
    const size_t o_st = take((size_t)N * 2 * heads * 4);
    if ((rc = ensure(c, &c->gws, &c->gws_bytes, off))) return rc;
    char* g = (char*)c->gws;
    float* st = (float*)(g + o_st);
    int32_t* node_graph = nullptr;   // (no node -> graph table on this path: both kernels walk graph_ptr on the scalar unit)
    const float* wa = p->buf;
    const unsigned* wx = reinterpret_cast<const unsigned*>(p->buf + (size_t)2 * heads * Fin);
    {
      ProfScope ps(c, s, "gat_stmax_kernel");
      HIPCHK(c, launch_gat_stmax(X, wa, N, Fin, heads, rowptr, col, graph_ptr, num_graphs, alpha, st, node_graph, gmax, c->gmax_cap, gen, s));
    }
    ProfScope ps(c, s, "gat_fused2_kernel");
    HIPCHK(c, launch_gat_fused(X, Fin, st, rowptr, col, graph_ptr, num_graphs, gmax, wx, N, heads, Fh, concat, alpha, out, c->gmax_cap, gen, s));
    return MGU_OK;
  }
  // gather path: Wh (N, HF) node table | st (N, 2H) attention scalars from ONE GEMM, then per-graph max, then the row gather
  const int NP = HF + 2 * heads;
  const size_t o_wh = take((size_t)N * HF * 4), o_st = take((size_t)N * 2 * heads * 4), o_ng = take((size_t)N * 4);
  if ((rc = ensure(c, &c->gws, &c->gws_bytes, off))) return rc;
  char* g = (char*)c->gws;
  float* Whp = (float*)(g + o_wh);
  float* st = (float*)(g + o_st);
  const mgu_gat_weights* pw = p;
  if (p->fused) return fail(c, MGU_ERR_STATE, "internal: aggregate-first weights on the gather path");
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = X;
  d.w = pw->buf;
  d.out = Whp;
  d.M = N, d.H = 1, d.W = N;
  d.Cp = Fin, d.ldin = Fin, d.KS = 1, d.K = Fin, d.Kp = pw->Kp;
  d.N = NP, d.ldout = HF;
  d.split_n = HF, d.out2 = st, d.ld2 = 2 * heads;
  {
    ProfScope ps(c, s, "igemm_kernel (GAT linear)");
    HIPCHK(c, launch_igemm_f32(d, s));
  }
  int32_t* node_graph = nullptr;  // node -> graph id (NULL: a single graph)
  if (num_graphs > 1) {
    node_graph = (int32_t*)(g + o_ng);
    HIPCHK(c, launch_gat_node_graph(graph_ptr, num_graphs, 0, N, node_graph, s));
  }
  {
    ProfScope ps(c, s, "gat_edge_max_kernel");
    HIPCHK(c, launch_gat_edge_max(st, rowptr, col, node_graph, N, heads, alpha, gmax, c->gmax_cap, gen, s));
  }
  ProfScope ps(c, s, "gat_aggregate_kernel");
  HIPCHK(c, launch_gat_aggregate(Whp, HF, st, rowptr, col, node_graph, gmax, N, E, heads, Fh, concat, alpha, out, c->gmax_cap, gen, s));
  return MGU_OK;
}

}  // namespace

extern "C" {

int mgu_gat_prepare(mgu_ctx* c, const void* W_dev, const void* a_dev, int heads, int Fout_head, int Fin, int has_edges,
                    mgu_gat_weights** out, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!W_dev || !a_dev || !out) return fail(c, MGU_ERR_INVALID, "NULL buffer");
  int rc = check_layer_shape(c, Fin, heads, Fout_head);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  mgu_gat_weights* p = new mgu_gat_weights();
  rc = prepare_into(c, p, (const float*)W_dev, (const float*)a_dev, heads, Fout_head, Fin, has_edges ? 1 : 0, (hipStream_t)hip_stream);
  if (rc) {
    if (p->buf) (void)hipFree(p->buf);
    delete p;
    return rc;
  }
  *out = p;
  return MGU_OK;
}

void mgu_gat_release(mgu_ctx* c, mgu_gat_weights* p) {
  if (!p) return;
  if (c) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (p->buf) (void)hipFree(p->buf);
  delete p;
}

int mgu_gat_layer_forward_prepared(mgu_ctx* c, const mgu_gat_weights* p, const void* X_dev, int N, const int32_t* rowptr_dev,
                                   const int32_t* col_dev, int64_t E, const int32_t* graph_ptr_dev, int num_graphs, int concat, float alpha,
                                   void* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!p || N < 0 || E < 0) return fail(c, MGU_ERR_INVALID, "bad GAT arguments");
  if (N == 0) return MGU_OK;
  if (!X_dev || !rowptr_dev || !out_dev || (E > 0 && !col_dev)) return fail(c, MGU_ERR_INVALID, "NULL buffer");
  if (p->fused && E == 0) return fail(c, MGU_ERR_INVALID, "weights were prepared with has_edges = 1 but the graph has no edges");
  HIPCHK(c, hipSetDevice(c->device));
  return forward_prepared(c, p, (const float*)X_dev, N, rowptr_dev, col_dev, E, graph_ptr_dev, num_graphs, concat, alpha, (float*)out_dev,
                          (hipStream_t)hip_stream);
}

// One-shot form: prepares the weights on every call (one more launch), then runs the prepared schedule.
int mgu_gat_layer_forward(mgu_ctx* c, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                          const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads,
                          int Fout_head, int concat, float alpha, void* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (N < 0 || E < 0) return fail(c, MGU_ERR_INVALID, "bad GAT arguments");
  int rc = check_layer_shape(c, Fin, heads, Fout_head);
  if (rc) return rc;
  if (N == 0) return MGU_OK;
  if (!X_dev || !rowptr_dev || !W_dev || !a_dev || !out_dev || (E > 0 && !col_dev)) return fail(c, MGU_ERR_INVALID, "NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  if (!c->gat_tmp) c->gat_tmp = new mgu_gat_weights();
  mgu_gat_weights* p = c->gat_tmp;
  const bool fused = c->tn.gat_fused && gat_fused_applicable(Fin, heads, Fout_head, E);
  const bool same_shape = p->buf && p->heads == heads && p->Fh == Fout_head && p->Fin == Fin && p->fused == fused;
  if (same_shape) {   // reuse the allocation, rebuild the contents (the weights may have changed)
    if (fused) {
      HIPCHK(c, launch_gat_prep((const float*)W_dev, (const float*)a_dev, p->buf, reinterpret_cast<unsigned*>(p->buf + (size_t)2 * heads * Fin), heads,
                                Fout_head, Fin, s));
    } else {
      HIPCHK(c, launch_pack_conv_w((const float*)W_dev, p->buf, 0, heads * Fout_head, Fin, Fin, 1, p->Kp, s));
      HIPCHK(c, launch_gat_wa_rows((const float*)W_dev, (const float*)a_dev, p->buf, heads * Fout_head, heads, Fout_head, Fin, p->Kp, s));
    }
  } else {
    if (p->buf) HIPCHK(c, hipDeviceSynchronize());   // an earlier call may still read the old allocation
    if ((rc = prepare_into(c, p, (const float*)W_dev, (const float*)a_dev, heads, Fout_head, Fin, E, s))) return rc;
  }
  return forward_prepared(c, p, (const float*)X_dev, N, rowptr_dev, col_dev, E, graph_ptr_dev, num_graphs, concat, alpha, (float*)out_dev, s);
}

}  // extern "C"

void mgud::gat_destroy(mgu_ctx* c) {
  if (c->gat_tmp) {
    if (c->gat_tmp->buf) (void)hipFree(c->gat_tmp->buf);
    delete c->gat_tmp;
    c->gat_tmp = nullptr;
  }
  if (c->gmaxbuf) (void)hipFree(c->gmaxbuf);
  c->gmaxbuf = nullptr;
}
