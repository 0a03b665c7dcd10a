// Training step schedule behind the C-ABI: train-mode forward (batch-statistics BatchNorm), backward
// (dgrad / wgrad / BN / ReLU / MaxPool / ConvTranspose / 1x1), softmax cross-entropy, Adam.
// Replaces the autograd graph the reference builds at scripts/train_segmentation.py:121-134
// (optimizer.zero_grad -> model(images) -> CrossEntropyLoss -> loss.backward -> optimizer.step).
// Host orchestration only; kernels live in igemm.hip, wgrad_f32.hip, train_kernels.hip.
#include <algorithm>

#include <string>

#include "ctx.h"
#include <cstdlib>

using namespace mgu;
using namespace mgud;

namespace {

struct TPlan {
  size_t xin = 0, bott = 0, ta = 0, tb = 0, tc = 0, dwp = 0, dwp_floats = 0, dgp = 0, dgp_floats = 0, wug = 0, sums = 0, total = 0;
  std::vector<size_t> z, y1, pooled, dcat;
};

TPlan plan_train(const mgu_ctx* c, int B, int H, int W) {
  const int d = c->depth;
  std::vector<int> hs, ws;
  level_dims(H, W, d, hs, ws);
  TPlan p;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  auto fl = [&](size_t n) { return take(n * sizeof(float)); };
  p.xin = fl((size_t)B * H * W * c->Cp0);
  p.z.assign(c->layers.size(), 0);
  p.y1.assign(c->layers.size(), 0);
  auto blk = [&](int li, int lvl) {
    const size_t M = (size_t)B * hs[lvl] * ws[lvl];
    const size_t C = c->layers[li].Cout;
    p.z[li] = fl(M * C);
    p.y1[li] = fl(M * C);
    p.z[li + 1] = fl(M * C);
  };
  for (int i = 0; i < d; ++i) blk(2 * i, i);
  blk(2 * d, d);
  for (int b = 0; b < d; ++b) blk(2 * d + 2 + 3 * b + 1, d - 1 - b);
  for (int i = 0; i < d; ++i) p.pooled.push_back(fl((size_t)B * hs[i + 1] * ws[i + 1] * ((size_t)c->feat << i)));
  p.bott = fl((size_t)B * hs[d] * ws[d] * ((size_t)c->feat << d));
  size_t tmax = 0;
  for (int i = 0; i <= d; ++i) tmax = std::max(tmax, (size_t)B * hs[i] * ws[i] * ((size_t)c->feat << i));
  p.ta = fl(tmax);
  p.tb = fl(tmax);
  p.tc = fl(tmax);
  for (int i = 0; i < d; ++i) p.dcat.push_back(fl((size_t)B * hs[i] * ws[i] * 2 * ((size_t)c->feat << i)));
  size_t pmax = (size_t)128 * 32;
  for (auto& L : c->layers) {
    pmax = std::max(pmax, (size_t)L.Np * L.Kp);                                              // wgrad panel (conv / convT)
    const size_t cop = rup(L.Cout, 4);
    pmax = std::max(pmax, (size_t)rup(L.Cp, 128) * rup(L.KS * L.KS * (int)cop * (L.convt ? 4 : 1), 32));  // dgrad panel
    if (L.convt) pmax = std::max(pmax, convt_x3_dgrad_floats(L.Cin, L.Cout));                          // its three-piece form
  }
  p.dwp_floats = std::max(pmax, (size_t)12 << 20);   // >= 48 MB: room for the atomics-free wgrad's partial panels
  p.dwp = fl(p.dwp_floats);
  p.dgp = fl(pmax);
  p.dgp_floats = pmax;
  size_t umax = 0;   // Winograd-transformed dgrad weights (conv_dgrad)
  for (const auto& L : c->layers)
    if (L.wino && rup(L.Cout, 4) % 16 == 0) umax = std::max(umax, wino_u_floats(L.Cin, rup(L.Cout, 4)));
  p.wug = fl(umax + 64);
  p.sums = take(sizeof(double) * 2 * ((size_t)c->feat << d) + 64);
  p.total = off;
  return p;
}

// The reduction slots are zero between launches: slot_reduce_kernel clears what it read, so only a fresh allocation
// needs a memset (instead of one before each of the ~60 per-channel reductions of a step).
int ensure_red(mgu_ctx* c) {
  const size_t need = chan_reduce_work_bytes(std::max(c->feat << c->depth, 64));
  if (c->redws_bytes >= need) return MGU_OK;
  int rc = ensure(c, &c->redws, &c->redws_bytes, need);
  if (rc) return rc;
  HIPCHK(c, hipMemset(c->redws, 0, need));
  return MGU_OK;
}

inline float* at(mgu_ctx* c, size_t off) { return (float*)((char*)c->tws + off); }

// conv (+bias) -> z ; batch statistics ; y = relu(bn(z)) written with pitch ldy
int conv_bn_relu_train(mgu_ctx* c, Layer& L, const float* in, int ldin, int B, int H, int W, float* z, float* y, int ldy,
                       double* sums, double* red, hipStream_t s, float* pooled = nullptr, bool* pool_done = nullptr) {
  const int64_t M = (int64_t)B * H * W;
  const int C = L.Cout;
  bool stats_done = false;   // Winograd layers accumulate sum z / sum z^2 in the conv epilogue
  int rc = run_layer(c, L, in, ldin, B, H, W, z, C, 0, 0, nullptr, L.b_src, 0, 0, s, nullptr, 0, nullptr, red, &stats_done);  // unet_encoder.py:16 / :20
  if (rc) return rc;
  if (stats_done) {
    HIPCHK(c, launch_bn_finalize_slots(red, c->last_stat_rows, sums, M, 1e-5f, 0.1f, L.gamma, L.beta, L.mean, L.invstd, L.tscale, L.tshift, L.run_mean,
                                       L.run_var, C, s));  // nn.BatchNorm2d defaults, unet_encoder.py:12-13
  } else {
    HIPCHK(c, launch_bn_stats(z, C, M, C, red, sums, s));
    HIPCHK(c, launch_bn_finalize(sums, sums + C, M, 1e-5f, 0.1f, L.gamma, L.beta, L.mean, L.invstd, L.tscale, L.tshift,
                                 L.run_mean, L.run_var, C, s));
  }
  if (pool_done) *pool_done = false;
  if (pooled && !(H & 1) && !(W & 1)) {   // MaxPool2d(2) of y in the same pass (even sizes: windows tile the image)
    HIPCHK(c, launch_bn_apply_relu_pool(z, L.tscale, L.tshift, y, ldy, pooled, B, H, W, C, s));
    if (pool_done) *pool_done = true;
  } else {
    HIPCHK(c, launch_bn_apply_relu(z, L.tscale, L.tshift, y, ldy, M, C, s));
  }
  L.t_in = in, L.t_ldin = ldin, L.t_z = z, L.t_y = y, L.t_ldy = ldy, L.t_B = B, L.t_H = H, L.t_W = W;
  return MGU_OK;
}

struct Bwd {
  mgu_ctx* c;
  hipStream_t s;
  float *ta, *tb, *dwp, *dgp, *wug, *flat;
  size_t dgp_floats = 0;
  size_t dwp_floats;
  double *sums, *red;
  int pend_rows = 0;   // rows of `red` holding the column sums of the last bn_relu_bwd (= the conv bias gradient), folded by the layer's
                       // gradient unpack launch (conv_wgrad)
};

// weight gradient of a conv layer into flat[off_w]: Z = dz (dense, pitch Cout), A = gather of the layer's input
int conv_wgrad(Bwd& w, const Layer& L, const float* dz) {
  mgu_ctx* c = w.c;
  WgradDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.z = dz, d.ldz = L.Cout, d.zoff = 0;
  d.in = L.t_in, d.ldin = L.t_ldin, d.inoff = 0, d.Cp = L.Cp;
  d.KS = L.KS;
  d.M = L.t_B * L.t_H * L.t_W, d.H = L.t_H, d.W = L.t_W;
  d.N = L.Cout, d.K = L.K, d.Kp = L.Kp;
  d.dw = w.dwp;
  d.dw_capacity = w.dwp_floats;
  {
    const double alg = 2.0 * d.M * (double)L.KS * L.KS * L.Cin * L.Cout;
    const bool ww = wino_wgrad_applicable(d) && !wgrad_thin_applicable(d);
    const bool x3 = ww && c->tn.wgrad_x3;   // three-piece mode: six bf16 MFMA products per Winograd multiply, on the bf16 pipe
    ProfScope ps(c, w.s,
                 wgrad_thin_applicable(d) ? "wgrad_thin kernels" : x3 ? "wino_wgrad_f32_kernel<X3>" : ww ? "wino_wgrad_f32_kernel" : "wgrad (direct) kernels",
                 alg, ww ? alg * 16.0 / 36.0 * (x3 ? 6.0 : 1.0) : alg, wgrad_thin_applicable(d) ? -1 : x3 ? 1 : 0);
    HIPCHK(c, launch_wgrad_f32(d, w.s));
  }
  // ... and, in the same launch, the fold of the bias gradient's column sums that bn_relu_bwd left in the reduction slots
  HIPCHK(c, launch_unpack_conv_grad(w.dwp, d.groups, (size_t)d.N * d.Kp, w.flat + L.off_w, L.Cout, L.Cin, L.Cp, L.KS, L.Kp, w.s, w.red,
                                    w.pend_rows, L.Cout, w.flat + L.off_b));
  w.pend_rows = 0;
  return MGU_OK;
}

// data gradient of a conv layer: din = conv(dz, flipped/transposed W) -> out (pitch ldout)
int conv_dgrad(Bwd& w, const Layer& L, const float* dz, float* out, int ldout) {
  mgu_ctx* c = w.c;
  const int Cop = rup(L.Cout, 4);
  const int Kd = L.KS * L.KS * Cop, Kpd = rup(Kd, 32);
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = dz, d.w = w.dgp, d.out = out;
  d.M = L.t_B * L.t_H * L.t_W, d.H = L.t_H, d.W = L.t_W;
  d.Cp = Cop, d.ldin = L.Cout == Cop ? L.Cout : Cop, d.KS = L.KS, d.K = Kd, d.Kp = Kpd;
  d.N = L.Cin, d.ldout = ldout;
  if (c->tn.wino_dgrad && L.wino && L.KS == 3 && Cop % 16 == 0 && c->tn.use_wino) {   // same Winograd kernel, weights flipped + transposed
    d.wu = L.wug ? L.wug : w.wug;
  }
  // only the weight form the chosen kernel reads is built: Winograd U (normally already packed with all the others by the last
  // weight refresh, repack_weights) or the direct flipped/transposed panel
  if (wino_applicable(d)) {
    if (!(L.wug && L.wug_valid)) {
      HIPCHK(c, launch_pack_wino_w(L.w_src, const_cast<float*>(d.wu), L.Cin, L.Cout, Cop, 1, c->tn.wino_prec, w.s));
      if (L.wug) L.wug_valid = true;
    }
  } else {
    HIPCHK(c, launch_pack_dgrad_w(L.w_src, w.dgp, L.Cout, L.Cin, Cop, L.KS, Kpd, w.s));
  }
  {
    const double alg = 2.0 * d.M * (double)L.KS * L.KS * L.Cin * L.Cout;
    const bool wn = wino_applicable(d);
    double mfma = 2.0 * d.M * (double)d.K * d.N;
    if (wn) mfma = 2.0 * L.t_B * ((L.t_H + 1) / 2) * ((L.t_W + 1) / 2) * 16.0 * d.Cp * d.N * (c->tn.wino_prec ? 6.0 : 1.0);
    // record under the name of the kernel family the dispatcher will pick (the same one as a forward conv of this shape)
    const std::string fam = std::string(igemm_kernel_name(d, 0));
    const char* label = "igemm/halo (dgrad)";
    if (fam == "wino3x3_cp_kernel<2>") label = "wino3x3_cp_kernel<2> (dgrad)";
    else if (fam == "wino3x3_cp_kernel<1>") label = "wino3x3_cp_kernel<1> (dgrad)";
    else if (wn) label = c->tn.wino_prec ? "wino3x3_f32_kernel<*,1> (dgrad)" : "wino3x3_f32_kernel<*,0> (dgrad)";
    ProfScope ps(c, w.s, label, alg, mfma, wn && c->tn.wino_prec ? 1 : 0);
    HIPCHK(c, launch_igemm_f32(d, w.s));
  }
  return MGU_OK;
}

// BN(train) + ReLU backward: dy (pitch lddy) -> dz (dense) ; fills dgamma, dbeta, conv bias grad
int bn_relu_bwd(Bwd& w, const Layer& L, const float* dy, int lddy, float* dz) {
  mgu_ctx* c = w.c;
  const int64_t M = (int64_t)L.t_B * L.t_H * L.t_W;
  const int C = L.Cout;
  HIPCHK(c, launch_bn_bwd_reduce(dy, lddy, L.tscale, L.tshift, L.t_z, C, L.mean, L.invstd, M, C, w.red, w.sums,
                                 w.flat + L.off_beta, w.flat + L.off_gamma, w.s));
  // dz and, fused, the column sums of dz (the conv bias gradient, analytically ~0 under BatchNorm): they stay in the reduction slots
  // until conv_wgrad of the SAME layer -- always the next user of the slots -- folds them inside its unpack launch
  HIPCHK(c, launch_bn_bwd_apply_deferred(dy, lddy, L.tscale, L.tshift, L.t_z, L.mean, L.invstd, L.gamma, w.sums, M, C, dz, w.red,
                                         &w.pend_rows, w.s));
  return MGU_OK;
}

// ConvBlock backward (unet_encoder.py:15-25 reversed).  dy has pitch lddy; dinput may be null.
int block_backward(Bwd& w, const Layer& L1, const Layer& L2, const float* dy, int lddy, float* dinput, int ld_dinput) {
  int rc;
  if ((rc = bn_relu_bwd(w, L2, dy, lddy, w.ta))) return rc;
  if ((rc = conv_wgrad(w, L2, w.ta))) return rc;
  if ((rc = conv_dgrad(w, L2, w.ta, w.tb, L2.Cin))) return rc;   // d(y1), dense pitch C
  if ((rc = bn_relu_bwd(w, L1, w.tb, L2.Cin, w.ta))) return rc;
  if ((rc = conv_wgrad(w, L1, w.ta))) return rc;
  if (dinput && (rc = conv_dgrad(w, L1, w.ta, dinput, ld_dinput))) return rc;
  return MGU_OK;
}

}  // namespace

size_t mgud::train_ws_bytes(const mgu_ctx* c, int B, int H, int W) { return plan_train(c, B, H, W).total; }

int mgud::unet_forward_train(mgu_ctx* c, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int64_t xs_w, int B,
                             int H, int W, float* logits, void* const* cat_dev, void* const* feat_dev, hipStream_t s) {
  const int d = c->depth;
  for (auto& L : c->layers)
    if (!L.w_src || !L.b_src || (!L.bn.empty() && (!L.gamma || !L.run_mean)))
      return fail(c, MGU_ERR_STATE, "training needs the parameter tensors recorded by mgu_unet_load_weights");
  const TPlan p = plan_train(c, B, H, W);
  c->have_train_fwd = false;
  c->want_train = true;    // from now on a weight refresh also packs the data-gradient Winograd sets (repack_weights)
  c->fold_dirty = true;    // this forward updates the BatchNorm running statistics in place: the eval fold is stale
  int rc = ensure(c, &c->tws, &c->tws_bytes, p.total);
  if (rc) return rc;
  if ((rc = ensure_red(c))) return rc;
  c->fold_dirty = true;   // this forward rewrites running_mean/var in place
  std::vector<int> hs, ws;
  level_dims(H, W, d, hs, ws);
  double* sums = (double*)((char*)c->tws + p.sums);
  double* red = (double*)c->redws;
  for (int i = 0; i < d; ++i)
    if (2 * hs[i + 1] != hs[i] || 2 * ws[i + 1] != ws[i])
      HIPCHK(c, hipMemsetAsync(cat_dev[i], 0, (size_t)B * hs[i] * ws[i] * 2 * ((size_t)c->feat << i) * sizeof(float), s));
  float* xin = at(c, p.xin);
  HIPCHK(c, launch_pack_input(x, xin, 0, B, c->in_ch, c->Cp0, H, W, xs_n, xs_c, xs_h, xs_w, s));

  c->t_cat.assign(d, nullptr);
  c->t_feat.assign(d, nullptr);
  c->t_pooled.assign(d, nullptr);
  const float* cur = xin;
  int ld = c->Cp0;
  for (int i = 0; i < d; ++i) {  // encoder
    const int C = c->feat << i, li = 2 * i;
    float* cat = (float*)cat_dev[i];
    if ((rc = conv_bn_relu_train(c, c->layers[li], cur, ld, B, hs[i], ws[i], at(c, p.z[li]), at(c, p.y1[li]), C, sums, red, s))) return rc;
    float* pooled = at(c, p.pooled[i]);
    bool pooled_done = false;
    if ((rc = conv_bn_relu_train(c, c->layers[li + 1], at(c, p.y1[li]), C, B, hs[i], ws[i], at(c, p.z[li + 1]), cat, 2 * C, sums, red, s, pooled,
                                 &pooled_done)))
      return rc;
    if (!pooled_done) HIPCHK(c, launch_maxpool2(cat, 2 * C, pooled, 0, B, hs[i], ws[i], C, s));
    c->t_cat[i] = cat, c->t_pooled[i] = pooled;
    cur = pooled, ld = C;
  }
  {  // bottleneck
    const int C = c->feat << d, li = 2 * d;
    float* bott = at(c, p.bott);
    if ((rc = conv_bn_relu_train(c, c->layers[li], cur, ld, B, hs[d], ws[d], at(c, p.z[li]), at(c, p.y1[li]), C, sums, red, s))) return rc;
    if ((rc = conv_bn_relu_train(c, c->layers[li + 1], at(c, p.y1[li]), C, B, hs[d], ws[d], at(c, p.z[li + 1]), bott, C, sums, red, s))) return rc;
    cur = bott, ld = C;
  }
  for (int b = 0; b < d; ++b) {  // decoder
    const int i = d - 1 - b, C = c->feat << i, lu = 2 * d + 2 + 3 * b;
    float* cat = (float*)cat_dev[i];
    float* feat = (float*)feat_dev[i];
    Layer& U = c->layers[lu];
    if ((rc = run_layer(c, U, cur, ld, B, hs[i + 1], ws[i + 1], cat, 2 * C, C, 0, nullptr, U.shift, hs[i], ws[i], s))) return rc;
    U.t_in = cur, U.t_ldin = ld, U.t_B = B, U.t_H = hs[i + 1], U.t_W = ws[i + 1];
    if ((rc = conv_bn_relu_train(c, c->layers[lu + 1], cat, 2 * C, B, hs[i], ws[i], at(c, p.z[lu + 1]), at(c, p.y1[lu + 1]), C, sums, red, s))) return rc;
    if ((rc = conv_bn_relu_train(c, c->layers[lu + 2], at(c, p.y1[lu + 1]), C, B, hs[i], ws[i], at(c, p.z[lu + 2]), feat, C, sums, red, s))) return rc;
    c->t_feat[i] = feat;
    cur = feat, ld = C;
  }
  Layer& F = c->layers.back();
  if (c->ncls <= 4) {
    HIPCHK(c, launch_conv1x1_head(cur, 0, ld, F.Cin, F.w_src, F.b_src, logits, c->ncls, c->ncls, (int64_t)B * H * W, s));
  } else if ((rc = run_layer(c, F, cur, ld, B, H, W, logits, c->ncls, 0, 0, nullptr, F.shift, 0, 0, s))) {
    return rc;
  }
  F.t_in = cur, F.t_ldin = ld, F.t_B = B, F.t_H = H, F.t_W = W;
  c->t_logits = logits;
  c->tB = B, c->tH = H, c->tW = W;
  c->have_train_fwd = true;
  return MGU_OK;
}

extern "C" {

// invalid data seen by an EARLIER kernel of this context (the word is host-mapped: no synchronisation needed to read it)
static int pending_data_error(mgu_ctx* c) {
  if (c->err_word && *(volatile int*)c->err_word) {
    const int w = *(volatile int*)c->err_word;
    *(volatile int*)c->err_word = 0;
    return fail(c, MGU_ERR_INVALID, "%s", err_word_message(w));
  }
  return MGU_OK;
}

int mgu_cross_entropy(mgu_ctx* c, const void* logits_dev, const int64_t* labels_dev, int64_t npix, int num_classes,
                      float grad_scale, void* dlogits_dev, float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!logits_dev || !labels_dev || !dlogits_dev || !loss_dev || npix < 1 || num_classes < 1)
    return fail(c, MGU_ERR_INVALID, "bad cross_entropy args");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = pending_data_error(c);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)hip_stream;
  if ((rc = ensure(c, &c->gws, &c->gws_bytes, 256))) return rc;
  if (!c->err_word) {
    HIPCHK(c, hipHostMalloc((void**)&c->err_word, sizeof(int), hipHostMallocMapped));
    *c->err_word = 0;
  }
  int* err_dev = nullptr;
  HIPCHK(c, hipHostGetDevicePointer((void**)&err_dev, c->err_word, 0));
  HIPCHK(c, launch_ce((const float*)logits_dev, labels_dev, npix, num_classes, -100 /* nn.CrossEntropyLoss default */, grad_scale,
                      (float*)dlogits_dev, rup(num_classes, 4), (double*)c->gws, err_dev, loss_dev, s));
  return MGU_OK;
}

int mgu_sync_check(mgu_ctx* c, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize((hipStream_t)hip_stream));
  return pending_data_error(c);
}

}  // extern "C"

// exchange != 0: the gradient of every finished block is mean-all-reduced on the context's communicator stream while the
// blocks below it are still being differentiated.  Blocks finish in reverse parameter order (final conv, decoder shallow ->
// deep, bottleneck, encoder deep -> shallow), so the finished part of the flat vector is a suffix that grows downwards;
// it is flushed whenever >= 4 MB are pending (xGMI collectives are latency-bound below that) and once at the end.
static int backward_body(mgu_ctx* c, const void* dlogits_dev, void* flat_grad_dev, void* hip_stream, int exchange) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->have_train_fwd) return fail(c, MGU_ERR_STATE, "mgu_unet_backward needs a preceding mgu_unet_forward(training=1)");
  if (!dlogits_dev || !flat_grad_dev) return fail(c, MGU_ERR_INVALID, "NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  {
    int prc = pending_data_error(c);
    if (prc) return prc;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  const int d = c->depth, B = c->tB, H = c->tH, W = c->tW;
  const TPlan p = plan_train(c, B, H, W);
  if (p.total > c->tws_bytes) return fail(c, MGU_ERR_STATE, "training workspace changed since the forward");
  std::vector<int> hs, ws;
  level_dims(H, W, d, hs, ws);
  Bwd w;
  w.c = c, w.s = s;
  w.ta = at(c, p.ta), w.tb = at(c, p.tb), w.dwp = at(c, p.dwp), w.dgp = at(c, p.dgp);
  w.dgp_floats = p.dgp_floats;
  w.dwp_floats = p.dwp_floats;
  w.wug = at(c, p.wug);
  w.flat = (float*)flat_grad_dev;
  w.sums = (double*)((char*)c->tws + p.sums);
  w.red = (double*)c->redws;
  float* tc = at(c, p.tc);
  int rc;
  int64_t pend_hi = c->nparams;   // flat[pend_lo, pend_hi) is finished but not yet exchanged
  auto block_done = [&](int64_t lo, bool last) -> int {
    if (!exchange) return MGU_OK;
    if (!last && (pend_hi - lo) < (1 << 20)) return MGU_OK;
    int r = comm_bucket(c, w.flat, lo, pend_hi, s);
    pend_hi = lo;
    return r;
  };

  // ---- final 1x1 conv (unet_decoder.py:143) ------------------------------------------------------
  const Layer& F = c->layers.back();
  const int64_t M0 = (int64_t)B * H * W;
  const int ldd = rup(c->ncls, 4), C0 = F.Cin;
  const float* dlog = (const float*)dlogits_dev;
  HIPCHK(c, launch_colsum(dlog, ldd, M0, ldd, w.red, (float*)w.sums, s));   // padded to ldd columns, then trimmed
  HIPCHK(c, hipMemcpyAsync(w.flat + F.off_b, w.sums, sizeof(float) * c->ncls, hipMemcpyDeviceToDevice, s));
  {
    WgradDesc g;
    memset(&g, 0, sizeof g);
    g.tn = &c->tn;
    g.z = dlog, g.ldz = ldd, g.in = F.t_in, g.ldin = F.t_ldin, g.Cp = C0, g.KS = 1;
    g.M = (int)M0, g.H = H, g.W = W, g.N = ldd, g.K = C0, g.Kp = F.Kp, g.dw = w.dwp, g.dw_capacity = w.dwp_floats;
    HIPCHK(c, launch_wgrad_f32(g, s));
    HIPCHK(c, launch_unpack_conv_grad(w.dwp, g.groups, (size_t)g.N * g.Kp, w.flat + F.off_w, c->ncls, C0, C0, 1, F.Kp, s));
    const int Kpd = rup(ldd, 32);
    // the panel of the data gradient: normally already packed with every other weight form by the last refresh (repack_weights)
    const float* dpanel = F.wxg_valid ? F.wxg : w.dgp;
    if (!F.wxg_valid) HIPCHK(c, launch_pack_dgrad_w(F.w_src, w.dgp, c->ncls, C0, ldd, 1, Kpd, s));
    IgemmDesc q;
    memset(&q, 0, sizeof q);
    q.tn = &c->tn;
    q.in = dlog, q.w = dpanel, q.out = tc, q.M = (int)M0, q.H = H, q.W = W, q.Cp = ldd, q.ldin = ldd, q.KS = 1, q.K = ldd,
    q.Kp = Kpd, q.N = C0, q.ldout = C0;
    HIPCHK(c, launch_igemm_f32(q, s));
  }
  const float* dy = tc;
  int lddy = C0;
  if ((rc = block_done(F.off_w, false))) return rc;

  // ---- decoder blocks, shallow -> deep (reverse of unet_decoder.py:139-141) ------------------------
  for (int b = d - 1; b >= 0; --b) {
    const int i = d - 1 - b, C = c->feat << i, lu = 2 * d + 2 + 3 * b;
    const Layer& U = c->layers[lu];
    float* dcat = at(c, p.dcat[i]);
    if ((rc = block_backward(w, c->layers[lu + 1], c->layers[lu + 2], dy, lddy, dcat, 2 * C))) return rc;
    // ConvTranspose2d backward (unet_decoder.py:36): d(up) = channels [C, 2C) of d(cat)
    const int64_t Mi = (int64_t)B * hs[i] * ws[i];
    if (2 * hs[i + 1] != hs[i] || 2 * ws[i + 1] != ws[i])  // F.pad backward (unet_decoder.py:46-47) drops the pad row/col
      HIPCHK(c, launch_zero_pad_region(dcat, 2 * C, C, C, B, hs[i], ws[i], 2 * hs[i + 1], 2 * ws[i + 1], s));
    HIPCHK(c, launch_colsum(dcat + C, 2 * C, Mi, C, w.red, w.flat + U.off_b, s));
    const int Kt = 4 * C, Kpt = rup(Kt, 32);
    {
      WgradDesc g;
      memset(&g, 0, sizeof g);
    g.tn = &c->tn;
      g.z = U.t_in, g.ldz = U.t_ldin, g.in = dcat, g.ldin = 2 * C, g.inoff = C, g.Cp = C, g.KS = 2;
      g.M = U.t_B * U.t_H * U.t_W, g.H = U.t_H, g.W = U.t_W, g.Hs = hs[i], g.Ws = ws[i];
      g.N = U.Cin, g.K = Kt, g.Kp = Kpt, g.dw = w.dwp, g.dw_capacity = w.dwp_floats;
      HIPCHK(c, launch_wgrad_f32(g, s));
      HIPCHK(c, launch_unpack_convt_grad(w.dwp, g.groups, (size_t)g.N * g.Kp, w.flat + U.off_w, U.Cin, C, Kpt, s));
    }
    IgemmDesc q;
    memset(&q, 0, sizeof q);
    q.tn = &c->tn;
    q.in = dcat + C, q.w = w.dgp, q.out = tc, q.M = U.t_B * U.t_H * U.t_W, q.H = U.t_H, q.W = U.t_W, q.Cp = C, q.ldin = 2 * C;
    q.KS = 2, q.K = Kt, q.Kp = Kpt, q.N = U.Cin, q.ldout = U.Cin, q.Hout = hs[i], q.Wout = ws[i];
    // three-piece kernel of the forward layer in its gather mode (convt_x3.hip) where the shapes allow, else the generic tile kernel
    q.wu = U.wxg_valid ? U.wxg : w.dgp;
    if (c->tn.convt_dgrad_x3 && convt_x3_dgrad_applicable(q) && (U.wxg_valid || convt_x3_dgrad_floats(U.Cin, C) <= w.dgp_floats)) {
      if (!U.wxg_valid) HIPCHK(c, launch_pack_convt_x3_dgrad(U.w_src, w.dgp, U.Cin, C, s));   // else: packed by the last weight refresh
    } else {
      q.wu = nullptr;
      HIPCHK(c, launch_pack_convt_dgrad_w(U.w_src, w.dgp, U.Cin, C, Kpt, s));
    }
    {
      const double alg = 2.0 * q.M * (double)Kt * U.Cin;
      ProfScope ps(c, s, q.wu ? "convt2x2_x3_kernel (dgrad)" : "igemm_kernel<f32> (ConvTranspose dgrad)", alg, q.wu ? 6.0 * alg : alg, q.wu ? 1 : 0);
      HIPCHK(c, launch_igemm_f32(q, s));
    }
    dy = tc, lddy = U.Cin;
    if ((rc = block_done(U.off_w, false))) return rc;
  }
  // ---- bottleneck -----------------------------------------------------------------------------------
  {
    const Layer& L1 = c->layers[2 * d];
    if ((rc = block_backward(w, L1, c->layers[2 * d + 1], dy, lddy, tc, L1.Cin))) return rc;
    if ((rc = block_done(L1.off_w, false))) return rc;
  }
  // ---- encoder blocks, deep -> shallow -----------------------------------------------------------------
  for (int i = d - 1; i >= 0; --i) {
    const int C = c->feat << i;
    float* dcat = at(c, p.dcat[i]);
    // d(skip) = d(cat)[:, :C] (decoder path) + MaxPool backward of d(pooled)
    HIPCHK(c, launch_maxpool2_bwd_add(c->t_cat[i], 2 * C, tc, dcat, 2 * C, B, hs[i], ws[i], C, s));
    const Layer& L1 = c->layers[2 * i];
    if ((rc = block_backward(w, L1, c->layers[2 * i + 1], dcat, 2 * C, i > 0 ? tc : nullptr, L1.Cin))) return rc;
    if ((rc = block_done(L1.off_w, i == 0))) return rc;
  }
  if (exchange && (rc = comm_join(c, s))) return rc;   // the caller's stream continues after the last bucket
  return MGU_OK;
}

static int backward_impl(mgu_ctx* c, const void* dlogits_dev, void* flat_grad_dev, void* hip_stream, int exchange) {
  const int rc = backward_body(c, dlogits_dev, flat_grad_dev, hip_stream, exchange);
  if (rc != MGU_OK && c && c->redws) {
    // an error return between a deferred column-sum pass and its fold would leave rows of the (otherwise self-cleaning) reduction
    // slots non-zero for every later reduction: clear them
    (void)hipMemsetAsync(c->redws, 0, c->redws_bytes, (hipStream_t)hip_stream);
  }
  if (rc != MGU_OK && exchange && c && c->comm) {
    // an error return after some buckets were issued: the caller's stream must still be ordered behind the communicator
    // stream (the buckets read and write flat_grad_dev), or the caller could free / reuse the buffer under a running collective
    const std::string keep = c->err;
    (void)comm_join(c, (hipStream_t)hip_stream);
    c->err = keep;
  }
  return rc;
}

extern "C" {

int mgu_unet_backward(mgu_ctx* c, const void* dlogits_dev, void* flat_grad_dev, void* hip_stream) {
  return backward_impl(c, dlogits_dev, flat_grad_dev, hip_stream, 0);
}

int mgu_unet_backward_allreduce(mgu_ctx* c, const void* dlogits_dev, void* flat_grad_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->comm) return fail(c, MGU_ERR_STATE, "mgu_unet_backward_allreduce needs mgu_comm_init_rank on this context");
  return backward_impl(c, dlogits_dev, flat_grad_dev, hip_stream, 1);
}

int mgu_adam_step(mgu_ctx* c, void* flat_param_dev, const void* flat_grad_dev, void* exp_avg_dev, void* exp_avg_sq_dev,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!flat_param_dev || !flat_grad_dev || !exp_avg_dev || !exp_avg_sq_dev || n < 0 || step < 1)
    return fail(c, MGU_ERR_INVALID, "bad adam args (step counts from 1)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_adam((float*)flat_param_dev, (const float*)flat_grad_dev, (float*)exp_avg_dev, (float*)exp_avg_sq_dev, n, lr,
                        beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)hip_stream));
  return MGU_OK;
}

int mgu_sgd_step(mgu_ctx* c, void* flat_param_dev, const void* flat_grad_dev, void* momentum_buf_dev, int64_t n, float lr,
                 float momentum, float weight_decay, int step, float grad_scale, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!flat_param_dev || !flat_grad_dev || (momentum != 0.f && !momentum_buf_dev) || n < 0 || step < 1 || momentum < 0.f)
    return fail(c, MGU_ERR_INVALID, "bad sgd args (step counts from 1; a momentum buffer when momentum != 0)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_sgd((float*)flat_param_dev, (const float*)flat_grad_dev, (float*)momentum_buf_dev, n, lr, momentum, weight_decay,
                       step, grad_scale, (hipStream_t)hip_stream));
  return MGU_OK;
}

}  // extern "C"
