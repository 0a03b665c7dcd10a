// C-ABI of libmgunet.so (see include/mgunet.h): context, weight repacking, U-Net forward schedule,
// GAT layer schedule.  Host orchestration only -- every arithmetic op is a kernel in igemm.hip,
// elementwise.hip or gat.hip.  No CPU fallback exists: without a HIP device every call fails.
#include "ctx.h"

#include <algorithm>

using namespace mgu;
using namespace mgud;

namespace mgud {
std::string g_create_err;
}

namespace {

struct WsPlan {
  size_t xin, tmp, bott, total;
  std::vector<size_t> pooled;
};

WsPlan plan_ws(const mgu_ctx* c, int B, int H, int W) {
  std::vector<int> hs, wsz;
  level_dims(H, W, c->depth, hs, wsz);
  WsPlan p;
  size_t off = 0;
  const size_t es = c->dtype == MGU_DTYPE_BF16 ? 2 : 4;
  auto take = [&](size_t elems) {
    size_t o = off;
    off += (elems * es + 255) / 256 * 256;
    return o;
  };
  p.xin = take((size_t)B * H * W * c->Cp0);
  size_t tmax = 0;
  for (int i = 0; i <= c->depth; ++i) {
    size_t f = (size_t)B * hs[i] * wsz[i] * ((size_t)c->feat << i);
    if (f > tmax) tmax = f;
  }
  p.tmp = take(tmax);
  for (int i = 0; i < c->depth; ++i) p.pooled.push_back(take((size_t)B * hs[i + 1] * wsz[i + 1] * ((size_t)c->feat << i)));
  p.bott = take((size_t)B * hs[c->depth] * wsz[c->depth] * ((size_t)c->feat << c->depth));
  p.total = off;
  return p;
}

}  // namespace

extern "C" {

const char* mgu_version(void) { return "mgunet 0.2 (gfx950: fp32 Winograd / bf16 MFMA forward, fp32 training step, GAT, RCCL gradient exchange)"; }

int mgu_create(int device_id, mgu_ctx** out) {
  if (!out) return fail(nullptr, MGU_ERR_INVALID, "out == NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, MGU_ERR_HIP, "no HIP device available (%s): libmgunet has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device_id < 0 || device_id >= n) return fail(nullptr, MGU_ERR_INVALID, "device %d out of range [0,%d)", device_id, n);
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(nullptr, MGU_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  mgu_ctx* c = new mgu_ctx();
  c->device = device_id;
  // kernel-selection switches live in the context (no process globals: two contexts never see each other's settings)
  auto flag = [](const char* name) { const char* v = getenv(name); return v && v[0] == '1'; };
  auto num = [](const char* name, int dflt) { const char* v = getenv(name); return v && v[0] ? atoi(v) : dflt; };
  Tuning& t = c->tn;
  t.first_mfma = !flag("MGU_NO_FIRST_MFMA");
  t.use_halo = !flag("MGU_NO_HALO");
  t.halo_tps3 = !flag("MGU_HALO_TPS1");
  t.halo_max_ppb = std::max(1, num("MGU_HALO_PPB", t.halo_max_ppb));
  t.use_wino = !flag("MGU_NO_WINOGRAD");
  t.wino_mode = num("MGU_WINO_MODE", -1);
  t.wino_prec = num("MGU_WINO_PREC", t.wino_prec) ? 1 : 0;
  t.wino_cp = !flag("MGU_NO_WINO_CP");
  t.wino_deep = !flag("MGU_NO_WINO_DEEP");
  t.convt_frag = !flag("MGU_NO_CONVT_FRAG");
  t.wino_yfast = flag("MGU_WINO_YFAST");
  t.wino_cp_narrow = num("MGU_WINO_CP_NARROW", 1) != 0;
  t.wino_rounds = std::max(1, num("MGU_WINO_ROUNDS", 1));
  t.wino_ppb_cap = std::max(1, num("MGU_WINO_PPB_CAP", 32));
  t.wgrad_halo = !flag("MGU_NO_WGRAD_HALO");
  t.wino_wgrad = !flag("MGU_NO_WINO_WGRAD");
  t.wgrad_x3 = !flag("MGU_NO_WGRAD_X3");
  t.convt_dgrad_x3 = !flag("MGU_NO_CONVT_DGRAD_X3");
  t.wgrad_thin = !flag("MGU_NO_THIN_WGRAD");
  t.wino_dgrad = !flag("MGU_NO_WINO_DGRAD");
  t.gat_fused = !flag("MGU_NO_GAT_FUSED");
  t.wino_ures = !flag("MGU_NO_WINO_URES");
  t.wino_prio = flag("MGU_WINO_PRIO");
  t.wino_asm = std::max(0, num("MGU_WINO_ASM", t.wino_asm));
  t.wino_asm_narrow = num("MGU_WINO_ASM_NARROW", 1) != 0;
  *out = c;
  return MGU_OK;
}

void mgu_destroy(mgu_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)mgu_comm_destroy(c);
  if (c->arena) (void)hipFree(c->arena);
  if (c->ws) (void)hipFree(c->ws);
  if (c->gws) (void)hipFree(c->gws);
  if (c->gbws) (void)hipFree(c->gbws);
  if (c->pack_dev) (void)hipFree(c->pack_dev);
  if (c->gbpanel) (void)hipFree(c->gbpanel);
  if (c->tws) (void)hipFree(c->tws);
  if (c->redws) (void)hipFree(c->redws);
  if (c->wuws) (void)hipFree(c->wuws);
  if (c->ncws) (void)hipFree(c->ncws);
  if (c->lossws) (void)hipFree(c->lossws);
  if (c->imgws) (void)hipFree(c->imgws);
  gat_destroy(c);
  if (c->err_word) (void)hipHostFree(c->err_word);
  for (auto e : c->ev) (void)hipEventDestroy(e);
  for (auto e : c->ev_total)
    if (e) (void)hipEventDestroy(e);
  delete c;
}

const char* mgu_last_error(mgu_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int mgu_unet_configure(mgu_ctx* c, int in_ch, int ncls, int feat, int depth, int dtype) {
  if (!c) return MGU_ERR_INVALID;
  if (in_ch < 1 || ncls < 1 || feat < 4 || (feat & 3) || depth < 1 || depth > 8)
    return fail(c, MGU_ERR_INVALID, "unsupported UNet(%d,%d,%d,%d): init_features must be a positive multiple of 4, depth 1..8",
                in_ch, ncls, feat, depth);
  if (dtype != MGU_DTYPE_F32 && dtype != MGU_DTYPE_BF16) return fail(c, MGU_ERR_INVALID, "unknown dtype %d", dtype);
  if (dtype == MGU_DTYPE_BF16 && ((feat & 7) || ncls > 4))
    return fail(c, MGU_ERR_INVALID, "bf16 storage needs init_features %% 8 == 0 (16-byte lanes of 8 bf16) and <= 4 classes");
  HIPCHK(c, hipSetDevice(c->device));
  c->in_ch = in_ch, c->ncls = ncls, c->feat = feat, c->depth = depth, c->dtype = dtype;
  c->Cp0 = rup(in_ch, dtype == MGU_DTYPE_BF16 ? 8 : 4);
  c->layers.clear();
  auto add_block = [&](const std::string& prefix, int cin, int cp, int cout) {  // ConvBlock, unet_encoder.py:4-25
    for (int j = 0; j < 2; ++j) {
      Layer L;
      L.prefix = prefix;
      L.conv = j == 0 ? "conv1" : "conv2";
      L.bn = j == 0 ? "bn1" : "bn2";
      L.Cin = j == 0 ? cin : cout;
      L.Cp = j == 0 ? cp : cout;
      L.Cout = cout;
      L.KS = 3;
      c->layers.push_back(L);
    }
  };
  int cin = in_ch, cp = c->Cp0, f = feat;
  for (int i = 0; i < depth; ++i) {  // unet_encoder.py:46-50
    add_block("encoder.encoder_blocks." + std::to_string(i) + ".", cin, cp, f);
    cin = cp = f;
    f *= 2;
  }
  add_block("encoder.bottleneck.", cin, cp, f);  // :53
  int prev = f;
  for (int b = 0; b < depth; ++b) {  // unet_decoder.py:103-114
    const int ci = feat << (depth - 1 - b);
    Layer U;
    U.prefix = "decoder.decoder_blocks." + std::to_string(b) + ".";
    U.conv = "upsample";
    U.Cin = U.Cp = prev;
    U.Cout = prev / 2;
    U.KS = 1;
    U.convt = true;
    c->layers.push_back(U);
    add_block(U.prefix + "conv_block.", ci + prev / 2, ci + prev / 2, ci);
    prev = ci;
  }
  Layer Fc;  // unet_decoder.py:117
  Fc.prefix = "decoder.";
  Fc.conv = "final_conv";
  Fc.Cin = Fc.Cp = prev;
  Fc.Cout = ncls;
  Fc.KS = 1;
  c->layers.push_back(Fc);

  size_t total = 0;
  for (auto& L : c->layers) {
    L.K = L.KS * L.KS * L.Cp;
    L.Kp = rup(L.K, dtype == MGU_DTYPE_BF16 ? 64 : 32);   // one 128-byte LDS row of k per pipeline step
    L.N = L.convt ? 4 * L.Cout : L.Cout;
    L.Np = rup(L.N, 128);
    total += (size_t)L.Np * L.Kp + 2 * (size_t)L.Np + (L.bn.empty() ? 0 : 4 * (size_t)L.Np);
    L.wino = dtype == MGU_DTYPE_F32 && !L.convt && L.KS == 3 && L.Cp % 16 == 0;   // Winograd F(2x2,3x3) layers (wino_f32.hip)
    if (L.wino) total += wino_u_floats(L.Cout, L.Cp);
    if (L.wino && rup(L.Cout, 4) % 16 == 0) total += wino_u_floats(L.Cin, rup(L.Cout, 4));   // data-gradient conv: roles swapped
    // fp32 ConvTranspose on fragment-ordered three-piece weights (convt_x3.hip).  (A bf16-storage sibling of that kernel -- one
    // fragment per operand straight from global memory -- was measured SLOWER than the LDS-tiled generic kernel, 0.178 vs 0.163 ms per
    // step: 32-byte row segments per K slice; not kept.)
    L.ctx3 = dtype == MGU_DTYPE_F32 && L.convt && L.Cin % 32 == 0 && L.Cout % 32 == 0 && c->tn.wino_prec != 0 && c->tn.convt_frag;
    if (L.ctx3) total += convt_x3_floats(L.Cin, L.Cout) + convt_x3_dgrad_floats(L.Cin, L.Cout);
    // bf16 storage: the layer's weights as bf16 MFMA fragments for convt2x2_bf16_kernel (whole-row LDS staging, transposed 16-byte stores)
    L.ctb = dtype == MGU_DTYPE_BF16 && L.convt && L.Cin % 64 == 0 && L.Cout % 32 == 0 && c->tn.convt_frag;
    if (L.ctb) total += convt_bf16f_floats(L.Cin, L.Cout);
    if (dtype == MGU_DTYPE_F32 && !L.convt && L.bn.empty()) total += (size_t)rup(L.Cin, 128) * rup(L.KS * L.KS * rup(L.Cout, 4), 32);   // final conv: data-gradient panel
    L.first = !L.convt && L.KS == 3 && first_conv_applicable(dtype, L.Cin, L.Cp, L.Cout, 8, 0);
    if (L.first) total += 9 * 4 * (size_t)L.Cout + first_mfma_floats();
  }
  // flat parameter order = the reference's named_parameters(): per ConvBlock conv1.{w,b}, conv2.{w,b},
  // bn1.{w,b}, bn2.{w,b} (unet_encoder.py:7-13); decoder block: upsample.{w,b} then its conv_block; final.
  {
    int64_t off = 0;
    size_t i = 0;
    auto wsize = [](const Layer& L) { return (int64_t)L.Cout * L.Cin * L.KS * L.KS * (L.convt ? 4 : 1); };
    while (i < c->layers.size()) {
      Layer& A = c->layers[i];
      if (!A.bn.empty()) {  // ConvBlock = two consecutive layers
        Layer& B2 = c->layers[i + 1];
        A.off_w = off, off += wsize(A);
        A.off_b = off, off += A.Cout;
        B2.off_w = off, off += wsize(B2);
        B2.off_b = off, off += B2.Cout;
        A.off_gamma = off, off += A.Cout;
        A.off_beta = off, off += A.Cout;
        B2.off_gamma = off, off += B2.Cout;
        B2.off_beta = off, off += B2.Cout;
        i += 2;
      } else {
        A.off_w = off, off += wsize(A);
        A.off_b = off, off += A.Cout;
        i += 1;
      }
    }
    c->nparams = off;
  }
  if (c->arena) HIPCHK(c, hipFree(c->arena));
  c->arena = nullptr;
  HIPCHK(c, hipMalloc((void**)&c->arena, total * sizeof(float)));
  HIPCHK(c, hipMemset(c->arena, 0, total * sizeof(float)));
  c->arena_floats = total;
  float* p = c->arena;
  for (auto& L : c->layers) {
    L.wp = p;
    p += (size_t)L.Np * L.Kp;
    L.scale = p;
    p += L.Np;
    L.shift = p;
    p += L.Np;
    L.wu = nullptr;
    if (L.wino) L.wu = p, p += wino_u_floats(L.Cout, L.Cp);
    L.wug = nullptr, L.wug_valid = false;
    if (L.wino && rup(L.Cout, 4) % 16 == 0) L.wug = p, p += wino_u_floats(L.Cin, rup(L.Cout, 4));
    if (L.ctx3) L.wu = p, p += convt_x3_floats(L.Cin, L.Cout);
    if (L.ctb) L.wu = p, p += convt_bf16f_floats(L.Cin, L.Cout);
    L.wxg = nullptr, L.wxg_valid = false;
    if (L.ctx3) L.wxg = p, p += convt_x3_dgrad_floats(L.Cin, L.Cout);
    if (c->dtype == MGU_DTYPE_F32 && !L.convt && L.bn.empty()) L.wxg = p, p += (size_t)rup(L.Cin, 128) * rup(L.KS * L.KS * rup(L.Cout, 4), 32);
    L.wf = nullptr;
    if (L.first) L.wf = p, p += 9 * 4 * (size_t)L.Cout;
    L.wfm = nullptr;
    if (L.first) L.wfm = p, p += first_mfma_floats();
    if (!L.bn.empty()) {
      L.mean = p, p += L.Np;
      L.invstd = p, p += L.Np;
      L.tscale = p, p += L.Np;
      L.tshift = p, p += L.Np;
    }
  }
  c->configured = true;
  c->have_train_fwd = false;
  c->loaded = false;
  return MGU_OK;
}

int64_t mgu_unet_param_count(mgu_ctx* c) {
  if (!c || !c->configured) return -1;
  return c->nparams;
}

int64_t mgu_unet_param_offset(mgu_ctx* c, const char* name) {
  if (!c || !c->configured || !name) return -1;
  const std::string key(name);
  for (auto& L : c->layers) {
    if (key == L.prefix + L.conv + ".weight") return L.off_w;
    if (key == L.prefix + L.conv + ".bias") return L.off_b;
    if (!L.bn.empty()) {
      if (key == L.prefix + L.bn + ".weight") return L.off_gamma;
      if (key == L.prefix + L.bn + ".bias") return L.off_beta;
    }
  }
  return -1;
}

int mgu_unet_load_weights(mgu_ctx* c, const mgu_tensor_desc* named, int n, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->configured) return fail(c, MGU_ERR_STATE, "mgu_unet_configure must precede mgu_unet_load_weights");
  if (!named || n <= 0) return fail(c, MGU_ERR_INVALID, "empty state_dict");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  std::map<std::string, const mgu_tensor_desc*> sd;
  for (int i = 0; i < n; ++i)
    if (named[i].name) sd[named[i].name] = &named[i];
  auto get = [&](const std::string& key, int64_t numel, const float** out) -> int {
    auto it = sd.find(key);
    if (it == sd.end()) return fail(c, MGU_ERR_INVALID, "state_dict is missing key '%s'", key.c_str());
    if (it->second->numel != numel)
      return fail(c, MGU_ERR_INVALID, "state_dict key '%s' has %lld elements, expected %lld", key.c_str(),
                  (long long)it->second->numel, (long long)numel);
    if (!it->second->ptr) return fail(c, MGU_ERR_INVALID, "state_dict key '%s' has a NULL pointer", key.c_str());
    *out = (const float*)it->second->ptr;
    return MGU_OK;
  };
  for (auto& L : c->layers) {
    const float *w, *b;
    const std::string cw = L.prefix + L.conv;
    int rc;
    if (L.convt) {
      if ((rc = get(cw + ".weight", (int64_t)L.Cin * L.Cout * 4, &w))) return rc;
      if ((rc = get(cw + ".bias", L.Cout, &b))) return rc;
      L.w_src = w, L.b_src = b;
    } else {
      if ((rc = get(cw + ".weight", (int64_t)L.Cout * L.Cin * L.KS * L.KS, &w))) return rc;
      if ((rc = get(cw + ".bias", L.Cout, &b))) return rc;
      L.w_src = w, L.b_src = b;
      if (!L.bn.empty()) {
        const float *g, *be, *rm, *rv;
        const std::string bn = L.prefix + L.bn;
        if ((rc = get(bn + ".weight", L.Cout, &g))) return rc;
        if ((rc = get(bn + ".bias", L.Cout, &be))) return rc;
        if ((rc = get(bn + ".running_mean", L.Cout, &rm))) return rc;
        if ((rc = get(bn + ".running_var", L.Cout, &rv))) return rc;
        L.gamma = g, L.beta = be, L.run_mean = const_cast<float*>(rm), L.run_var = const_cast<float*>(rv);
      }
    }
  }
  int rc = repack_weights(c, s);
  if (rc) return rc;
  c->loaded = true;
  return MGU_OK;
}

int mgu_unet_refresh_weights(mgu_ctx* c, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->configured || !c->loaded) return fail(c, MGU_ERR_STATE, "mgu_unet_refresh_weights needs a preceding mgu_unet_load_weights");
  HIPCHK(c, hipSetDevice(c->device));
  return repack_weights(c, (hipStream_t)hip_stream);
}

int mgu_unet_workspace_bytes(mgu_ctx* c, int B, int H, int W, int training, size_t* out) {
  if (!c || !out) return MGU_ERR_INVALID;
  if (!c->configured) return fail(c, MGU_ERR_STATE, "not configured");
  // eval: packed input, one conv1 temp, pooled tensors, bottleneck; training: every layer's z / y kept for backward,
  // gradient temporaries and the weight-gradient partial panels (mgunet_train.hip)
  *out = training ? train_ws_bytes(c, B, H, W) : plan_ws(c, B, H, W).total;
  return MGU_OK;
}

int mgu_unet_reserve(mgu_ctx* c, int B, int H, int W, int training) {
  size_t need = 0;
  int rc = mgu_unet_workspace_bytes(c, B, H, W, training, &need);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  return training ? ensure(c, &c->tws, &c->tws_bytes, need) : ensure(c, &c->ws, &c->ws_bytes, need);
}

}  // extern "C"

// Every packed weight form from the parameter tensors recorded by mgu_unet_load_weights (their CONTENTS may have changed: an
// optimizer step through the flat buffer).  All Winograd sets -- and, once the context has trained, the data-gradient sets --
// go out in one launch; the eval BatchNorm fold stays lazy.
int mgud::repack_weights(mgu_ctx* c, hipStream_t s) {
  std::vector<WinoPackItem> items;   // every form goes out in the one pack_wino_w_multi_kernel launch (kinds: common.h)
  for (auto& L : c->layers) {
    const float *w = L.w_src, *b = L.b_src;
    L.wxg_valid = false;
    if (L.convt) {
      // the generic panel is read by the fallback kernels only: built on first use when the layer runs on its three-piece fragments
      L.wp_dirty = L.ctx3;
      if (!L.ctx3) HIPCHK(c, launch_pack_convt_w(w, L.wp, c->dtype, L.Cin, L.Cout, L.Kp, s));
      if (L.ctb) HIPCHK(c, launch_pack_convt_bf16f(w, L.wu, L.Cin, L.Cout, s));
      if (L.ctx3) items.push_back(WinoPackItem{w, L.wu, L.Cout, L.Cin, 0, 0, 0, 0, PACK_CONVT_X3});
      items.push_back(WinoPackItem{b, L.shift, L.Cout, 4, 0, 0, 0, 0, PACK_BIAS_TILE});   // scale unused (nullptr at launch)
      if (L.wxg && c->want_train && c->tn.convt_dgrad_x3 && (L.Cin & 63) == 0 && (L.Cout & 31) == 0) {
        items.push_back(WinoPackItem{w, L.wxg, L.Cout, L.Cin, 0, 0, 1, 0, PACK_CONVT_X3});
        L.wxg_valid = true;
      }
    } else {
      // a layer that runs as Winograd / first-conv / 1x1 head kernel reads wu / wf / w_src; its direct panel is packed lazily, only
      // if a launch ever falls back to the implicit-GEMM kernel (run_layer)
      L.wp_dirty = true;
      if (L.wu) items.push_back(WinoPackItem{w, L.wu, L.Cout, L.Cin, L.Cp, 0, 0, 0, PACK_WINO});
      L.wug_valid = false;
      if (L.wug && c->want_train && c->tn.wino_dgrad && c->tn.use_wino) {
        items.push_back(WinoPackItem{w, L.wug, L.Cin, L.Cout, rup(L.Cout, 4), 0, 1, 0, PACK_WINO});
        L.wug_valid = true;
      }
      const bool head_kernel = L.bn.empty() && c->ncls <= 4;   // conv1x1_head_kernel reads the reference's (ncls, C) weight itself
      if (!L.wu && !L.first && !head_kernel) {
        HIPCHK(c, launch_pack_conv_w(w, L.wp, c->dtype, L.Cout, L.Cin, L.Cp, L.KS, L.Kp, s));
        L.wp_dirty = false;
      }
      if (L.wf) items.push_back(WinoPackItem{w, L.wf, L.Cout, L.Cin, 0, 0, 0, 0, PACK_FIRST_W});
      if (L.wfm && L.Cout == 32 && L.Cin <= 3) items.push_back(WinoPackItem{w, L.wfm, L.Cout, L.Cin, 0, 0, 0, 0, PACK_FIRST_MFMA});
      if (!L.bn.empty()) c->fold_dirty = true;   // eval scale/shift are folded lazily by the next eval forward (training never reads them)
      else items.push_back(WinoPackItem{b, L.shift, L.Cout, 1, 0, 0, 0, 0, PACK_BIAS_TILE});
      if (L.bn.empty() && L.wxg && c->want_train) {   // final conv: panel of its data gradient (mgu_unet_backward)
        const int Cop = rup(L.Cout, 4);
        items.push_back(WinoPackItem{w, L.wxg, L.Cout, L.Cin, Cop, rup(L.KS * L.KS * Cop, 32), L.KS, 0, PACK_DGRAD_W});
        L.wxg_valid = true;
      }
    }
  }
  // batches of <= WINO_PACK_MAX items; the tables are uploaded only when they differ from what the device already holds (a
  // refresh after an optimizer step finds them unchanged: same tensors, same buffers)
  std::vector<WinoPackBatch> batches;
  for (size_t i0 = 0; i0 < items.size(); i0 += WINO_PACK_MAX) {
    WinoPackBatch b;
    memset(&b, 0, sizeof b);
    b.n = (int)std::min<size_t>(WINO_PACK_MAX, items.size() - i0), b.prec = c->tn.wino_prec;
    for (int i = 0; i < b.n; ++i) b.it[i] = items[i0 + i];
    if (!wino_pack_batch_prepare(b)) return fail(c, MGU_ERR_STATE, "internal: a Winograd layer is not packable");
    batches.push_back(b);
  }
  const bool same = batches.size() == c->pack_host.size() &&
                    (batches.empty() || memcmp(batches.data(), c->pack_host.data(), batches.size() * sizeof(WinoPackBatch)) == 0);
  if (!same) {
    if ((int)batches.size() > c->pack_dev_cap) {
      if (c->pack_dev) HIPCHK(c, hipFree(c->pack_dev));
      c->pack_dev = nullptr;
      HIPCHK(c, hipMalloc((void**)&c->pack_dev, batches.size() * sizeof(WinoPackBatch)));
      c->pack_dev_cap = (int)batches.size();
    }
    HIPCHK(c, hipStreamSynchronize(s));   // an earlier launch may still read the old table
    HIPCHK(c, hipMemcpy(c->pack_dev, batches.data(), batches.size() * sizeof(WinoPackBatch), hipMemcpyHostToDevice));
    c->pack_host = batches;
  }
  for (size_t k = 0; k < batches.size(); ++k) HIPCHK(c, launch_pack_wino_w_multi(c->pack_dev + k, batches[k].total_blocks, s));
  return MGU_OK;
}

int mgud::run_layer(mgu_ctx* c, const Layer& L, const void* in_v, int ldin, int B, int H, int W, void* out_v, int ldout,
                    int coff, int relu, const float* scale, const float* shift, int Hout, int Wout, hipStream_t s,
                    void* pool, int ldpool, bool* pool_fused, double* stat_slots, bool* stat_fused) {
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = (const float*)in_v;   // element type follows c->dtype; the descriptor carries raw pointers
  d.w = L.wp;
  d.wu = L.wu;
  d.scale = scale;
  d.shift = shift;
  d.out = (float*)out_v;
  d.M = B * H * W;
  d.H = H;
  d.W = W;
  d.Cp = L.Cp;
  d.ldin = ldin;
  d.KS = L.KS;
  d.K = L.K;
  d.Kp = L.Kp;
  d.N = L.N;
  d.ldout = ldout;
  d.coff = coff;
  d.relu = relu;
  d.out_mode = L.convt ? 1 : 0;
  d.ct_cout = L.Cout;
  d.Hout = Hout;
  d.Wout = Wout;
  if (pool_fused) *pool_fused = false;
  if (stat_fused) *stat_fused = false;
  if (L.wfm && c->tn.first_mfma && ldin == L.Cp && first_mfma_applicable(c->dtype, L.Cin, L.Cp, L.Cout, ldout, coff, H, W)) {
    const double alg = 2.0 * d.M * 9.0 * L.Cin * L.Cout;
    ProfScope ps(c, s, "conv3x3_first_mfma_kernel", alg, 2.0 * d.M * 32.0 * 32.0 * (c->dtype == MGU_DTYPE_F32 ? 6.0 : 3.0), 1);
    HIPCHK(c, launch_first_mfma(c->dtype, in_v, L.wfm, scale, shift, out_v, B, H, W, ldout, coff, relu, s));
    return MGU_OK;
  }
  if (L.wf && ldin == L.Cp && first_conv_applicable(c->dtype, L.Cin, L.Cp, L.Cout, ldout, coff) &&
      (int64_t)B * H * W * std::max(ldout, 8) < (1ll << 31)) {
    ProfScope ps(c, s, "conv3x3_first_kernel", 2.0 * d.M * 9.0 * L.Cin * L.Cout, 0, -1);
    HIPCHK(c, launch_first_conv(c->dtype, in_v, L.wf, scale, shift, out_v, B, H, W, L.Cin, L.Cout, ldout, coff, relu, s));
    return MGU_OK;
  }
  // the Winograd epilogue also accumulates sum z, sum z^2: one accumulator row per workgroup, so only while the launch's grid fits
  // the table (>= 19 images of 512^2 or 5 of 1024^2 per GPU on the full-resolution 32-channel layer, or a small MGU_WINO_PPB_CAP, do
  // not: the caller then takes the separate statistics pass, launch_bn_stats)
  if (stat_slots && c->dtype == MGU_DTYPE_F32 && wino_applicable(d) && wino_grid_blocks(d) <= STAT_ROWS) {
    d.stat_slots = stat_slots;
    c->last_stat_rows = wino_grid_blocks(d);
    if (stat_fused) *stat_fused = true;
  }
  if (pool && ((c->dtype == MGU_DTYPE_F32 && wino_applicable(d)) || halo_pool_fusable(d, c->dtype))) {
    // the Winograd / halo epilogue also writes the 2x2 max-pooled tensor
    d.pool = (float*)pool, d.ldpool = ldpool;
    if (pool_fused) *pool_fused = true;
  }
  if (L.wp_dirty) {   // falling back to the direct kernel: build its panel now
    const bool direct = L.convt ? !(c->dtype == MGU_DTYPE_F32 && convt_x3_applicable(d)) : !(c->dtype == MGU_DTYPE_F32 && wino_applicable(d));
    if (direct) {
      if (L.convt) HIPCHK(c, launch_pack_convt_w(L.w_src, L.wp, c->dtype, L.Cin, L.Cout, L.Kp, s));
      else HIPCHK(c, launch_pack_conv_w(L.w_src, L.wp, c->dtype, L.Cout, L.Cin, L.Cp, L.KS, L.Kp, s));
      L.wp_dirty = false;
    }
  }
  // profiling record: algorithmic 2*MAC of the operator and what the matrix pipe really issues (Winograd F(2x2,3x3): 16 products
  // per 2x2 tile and channel pair; the three-piece operand split issues six bf16 products per fp32 product)
  const double alg = L.convt ? 2.0 * d.M * (double)L.Cin * L.Cout * 4.0 : 2.0 * d.M * (double)L.KS * L.KS * L.Cin * L.Cout;
  double mfma = L.convt ? alg : 2.0 * d.M * (double)L.K * L.Cout;
  int pipe = c->dtype == MGU_DTYPE_BF16 ? 1 : 0;
  if (c->dtype == MGU_DTYPE_F32 && convt_x3_applicable(d)) mfma = 6.0 * alg, pipe = 1;
  if (c->dtype == MGU_DTYPE_F32 && wino_applicable(d)) {
    mfma = 2.0 * B * ((H + 1) / 2) * ((W + 1) / 2) * 16.0 * L.Cp * L.Cout * (c->tn.wino_prec ? 6.0 : 1.0);
    pipe = c->tn.wino_prec ? 1 : 0;
  }
  ProfScope ps(c, s, igemm_kernel_name(d, c->dtype), alg, mfma, pipe);
  if (c->dtype == MGU_DTYPE_BF16) HIPCHK(c, launch_igemm_bf16(d, s));
  else HIPCHK(c, launch_igemm_f32(d, s));
  return MGU_OK;
}

static int run_conv(mgu_ctx* c, const Layer& L, const void* in, int ldin, int B, int H, int W, void* out, int ldout,
                    int coff, int relu, int Hout, int Wout, hipStream_t s, void* pool = nullptr, int ldpool = 0,
                    bool* pool_fused = nullptr) {  // eval: folded BN scale/shift
  return run_layer(c, L, in, ldin, B, H, W, out, ldout, coff, relu, L.bn.empty() ? nullptr : L.scale, L.shift, Hout, Wout, s,
                   pool, ldpool, pool_fused, nullptr, nullptr);
}

extern "C" {

int mgu_unet_forward(mgu_ctx* c, const void* x_dev, int B, int H, int W, int64_t xs_n, int64_t xs_c, int64_t xs_h,
                     int64_t xs_w, void* logits_dev, void* const* cat_dev, void* const* feat_dev, int training,
                     void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->configured || !c->loaded) return fail(c, MGU_ERR_STATE, "configure + load_weights must precede forward");
  if (!x_dev || !logits_dev || !cat_dev || !feat_dev) return fail(c, MGU_ERR_INVALID, "NULL buffer");
  const int depth = c->depth;
  if (B < 1 || H < (1 << depth) || W < (1 << depth))
    return fail(c, MGU_ERR_INVALID, "input %dx%dx%d too small for depth %d", B, H, W, depth);
  if ((int64_t)B * H * W >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "B*H*W must be < 2^31");
  for (int i = 0; i < depth; ++i)
    if (!cat_dev[i] || !feat_dev[i]) return fail(c, MGU_ERR_INVALID, "NULL cat/feat buffer %d", i);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  if (training && c->dtype != MGU_DTYPE_F32)
    return fail(c, MGU_ERR_STATE, "training runs in fp32 only (bf16 storage is an inference mode)");
  if (training) {  // batch-statistics BatchNorm, running-stat update, activations kept for mgu_unet_backward
    int rc = unet_forward_train(c, (const float*)x_dev, xs_n, xs_c, xs_h, xs_w, B, H, W, (float*)logits_dev, cat_dev, feat_dev, s);
    if (rc == MGU_OK && c->pm_out) {   // a pending patch-mean request is served by the stand-alone kernel
      HIPCHK(c, launch_patch_mean(feat_dev[0], 0, (float*)c->pm_out, B, H, W, c->feat, c->pm_patch, s));
      c->pm_out = nullptr;
    }
    return rc;
  }
  const WsPlan plan = plan_ws(c, B, H, W);
  int rc = ensure(c, &c->ws, &c->ws_bytes, plan.total);
  if (rc) return rc;
  if (c->fold_dirty) {  // eval BatchNorm: y = scale * conv + shift with the CURRENT running statistics
    for (auto& L : c->layers)
      if (!L.bn.empty()) HIPCHK(c, launch_bn_fold(L.b_src, L.gamma, L.beta, L.run_mean, L.run_var, 1e-5f, L.scale, L.shift, L.Cout, s));
    c->fold_dirty = false;
  }
  char* ws = (char*)c->ws;
  const size_t es = c->dtype == MGU_DTYPE_BF16 ? 2 : 4;
  void* xin = ws + plan.xin;
  void* tmp = ws + plan.tmp;
  void* bott = ws + plan.bott;
  std::vector<int> hs, wsz;
  level_dims(H, W, depth, hs, wsz);

  if (c->prof) {
    for (auto& e : c->ev_total)
      if (!e) HIPCHK(c, hipEventCreate(&e));
    HIPCHK(c, hipEventRecord(c->ev_total[0], s));
  }

  // odd sizes: F.pad (unet_decoder.py:46-47) leaves a zero row/column in the up-sampled half
  for (int i = 0; i < depth; ++i)
    if (2 * hs[i + 1] != hs[i] || 2 * wsz[i + 1] != wsz[i])
      HIPCHK(c, hipMemsetAsync(cat_dev[i], 0, (size_t)B * hs[i] * wsz[i] * 2 * ((size_t)c->feat << i) * es, s));

  // the first convolution on the matrix cores reads the caller's image itself (first_mfma.hip): no packed copy of the input
  const Layer& L0 = c->layers[0];
  const bool first_direct = L0.wfm && c->tn.first_mfma && first_mfma_applicable(c->dtype, L0.Cin, L0.Cp, L0.Cout, c->feat, 0, H, W) &&
                            (int64_t)B * H * W < (1ll << 31);
  if (!first_direct) HIPCHK(c, launch_pack_input((const float*)x_dev, xin, c->dtype, B, c->in_ch, c->Cp0, H, W, xs_n, xs_c, xs_h, xs_w, s));

  int li = 0;
  const void* cur = xin;
  int cur_ld = c->Cp0;
  for (int i = 0; i < depth; ++i) {  // encoder, unet_encoder.py:67-70
    const int C = c->feat << i;
    if (i == 0 && first_direct) {
      if (c->fold_dirty) return fail(c, MGU_ERR_STATE, "internal: eval scale/shift not folded");
      const double alg = 2.0 * B * H * W * 9.0 * L0.Cin * L0.Cout;
      ProfScope ps(c, s, "conv3x3_first_mfma_kernel", alg, 2.0 * B * H * W * 32.0 * 32.0 * (c->dtype == MGU_DTYPE_F32 ? 6.0 : 3.0), 1);
      HIPCHK(c, launch_first_mfma_direct(c->dtype, (const float*)x_dev, xs_n, xs_c, xs_h, xs_w, c->in_ch, L0.wfm, L0.bn.empty() ? nullptr : L0.scale,
                                         L0.shift, tmp, B, H, W, C, 0, 1, s));
      ++li;
    } else if ((rc = run_conv(c, c->layers[li++], cur, cur_ld, B, hs[i], wsz[i], tmp, C, 0, 1, 0, 0, s))) return rc;
    void* pooled = ws + plan.pooled[i];
    bool fused = false;   // MaxPool2d(2) (unet_encoder.py:48) rides in the conv2 epilogue on the Winograd path
    if ((rc = run_conv(c, c->layers[li++], tmp, C, B, hs[i], wsz[i], cat_dev[i], 2 * C, 0, 1, 0, 0, s, pooled, C, &fused))) return rc;
    if (!fused) HIPCHK(c, launch_maxpool2(cat_dev[i], 2 * C, pooled, c->dtype, B, hs[i], wsz[i], C, s));
    cur = pooled;
    cur_ld = C;
  }
  {  // bottleneck, :72
    const int C = c->feat << depth;
    if ((rc = run_conv(c, c->layers[li++], cur, cur_ld, B, hs[depth], wsz[depth], tmp, C, 0, 1, 0, 0, s))) return rc;
    if ((rc = run_conv(c, c->layers[li++], tmp, C, B, hs[depth], wsz[depth], bott, C, 0, 1, 0, 0, s))) return rc;
    cur = bott;
    cur_ld = C;
  }
  for (int b = 0; b < depth; ++b) {  // decoder, unet_decoder.py:139-141
    const int i = depth - 1 - b;
    const int C = c->feat << i;
    // ConvTranspose2d(k2,s2) -> pixel-shuffle store into channels [C, 2C) of the concat buffer (:36,:53)
    if ((rc = run_conv(c, c->layers[li++], cur, cur_ld, B, hs[i + 1], wsz[i + 1], cat_dev[i], 2 * C, C, 0, hs[i], wsz[i], s)))
      return rc;
    if ((rc = run_conv(c, c->layers[li++], cat_dev[i], 2 * C, B, hs[i], wsz[i], tmp, C, 0, 1, 0, 0, s))) return rc;
    if ((rc = run_conv(c, c->layers[li++], tmp, C, B, hs[i], wsz[i], feat_dev[i], C, 0, 1, 0, 0, s))) return rc;
    cur = feat_dev[i];
    cur_ld = C;
  }
  // final 1x1 conv (:143): a few output channels -> HBM-bound head kernel reading the reference's (ncls, C) weight
  {
    const Layer& F = c->layers[li++];
    const int pm_dtype = c->dtype;   // decoder features are stored in the compute dtype
    if (c->pm_out && F.w_src && F.b_src && patch_mean_head_fusable(pm_dtype, F.Cin, c->ncls) && F.Cin <= 256) {
      // requested patch means + the 1x1 head in ONE pass over the decoder feature (both are pure bandwidth)
      ProfScope ps(c, s, "patch_mean_kernel (1x1 head + patch means)", 2.0 * B * H * W * F.Cin * c->ncls, 0, -1);
      HIPCHK(c, launch_patch_mean(cur, pm_dtype, (float*)c->pm_out, B, H, W, F.Cin, c->pm_patch, s, F.w_src, F.b_src,
                                  (float*)logits_dev, c->ncls));
      c->pm_out = nullptr;
    } else if (c->ncls <= 4 && F.w_src && F.b_src) {
      ProfScope ps(c, s, "conv1x1_head_kernel", 2.0 * B * H * W * F.Cin * c->ncls, 0, -1);
      HIPCHK(c, launch_conv1x1_head(cur, c->dtype, cur_ld, F.Cin, F.w_src, F.b_src, (float*)logits_dev, c->ncls, c->ncls,
                                    (int64_t)B * H * W, s));   // logits are always fp32
    } else if ((rc = run_conv(c, F, cur, cur_ld, B, H, W, logits_dev, c->ncls, 0, 0, 0, 0, s))) {
      return rc;
    }
  }
  if (c->pm_out) {   // request not served by the fused pass (head shape): separate kernel, same result
    HIPCHK(c, launch_patch_mean(cur, c->dtype, (float*)c->pm_out, B, H, W, c->layers.back().Cin, c->pm_patch, s));
    c->pm_out = nullptr;
  }
  if (c->prof) HIPCHK(c, hipEventRecord(c->ev_total[1], s));
  return MGU_OK;
}

int mgu_unet_request_patch_mean(mgu_ctx* c, int patch, void* out_dev) {
  if (!c) return MGU_ERR_INVALID;
  if (!c->configured) return fail(c, MGU_ERR_STATE, "not configured");
  if (!out_dev) {   // cancel a pending request
    c->pm_out = nullptr;
    return MGU_OK;
  }
  if (patch < 1) return fail(c, MGU_ERR_INVALID, "bad patch_mean request");
  const int C = c->layers.back().Cin, vec = c->dtype == MGU_DTYPE_BF16 ? 8 : 4;
  if ((C % vec) || C > 256) return fail(c, MGU_ERR_INVALID, "patch means need init_features %% %d == 0 and <= 256 (got %d)", vec, C);
  c->pm_patch = patch;
  c->pm_out = out_dev;
  return MGU_OK;
}

// scratch for the building-block entry points: packed panel + scale/shift, grown on demand
static int block_scratch(mgu_ctx* c, int Np, int Kp, float** wp, float** scale, float** shift, hipStream_t s) {
  const size_t need = ((size_t)Np * Kp + 2 * (size_t)Np) * sizeof(float);
  int rc = ensure(c, &c->gws, &c->gws_bytes, need);
  if (rc) return rc;
  *wp = (float*)c->gws;
  *scale = *wp + (size_t)Np * Kp;
  *shift = *scale + Np;
  HIPCHK(c, hipMemsetAsync(c->gws, 0, need, s));
  return MGU_OK;
}

// Packed forms of one Conv2d weight (direct panel and, for fp32 3x3 layers with Cin % 16 == 0, the Winograd U), owned by the
// library: built once by mgu_conv2d_prepare and reused by every mgu_conv2d_prepared_nhwc call until the weight changes.
struct mgu_conv_weights {
  int Cout = 0, Cin = 0, ksize = 0, K = 0, Kp = 0, Np = 0;
  float *wp = nullptr, *wu = nullptr, *shift = nullptr;   // one allocation: [panel | bias-as-shift | U]
};

int mgu_conv2d_prepare(mgu_ctx* c, const void* w_dev, int Cout, int Cin, int ksize, mgu_conv_weights** out, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!w_dev || !out || Cout < 1 || (ksize != 1 && ksize != 3)) return fail(c, MGU_ERR_INVALID, "bad conv2d_prepare args (ksize must be 1 or 3)");
  if (Cin < 4 || (Cin & 3)) return fail(c, MGU_ERR_INVALID, "conv2d needs Cin %% 4 == 0 (got %d)", Cin);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  mgu_conv_weights* p = new mgu_conv_weights();
  p->Cout = Cout, p->Cin = Cin, p->ksize = ksize;
  p->K = ksize * ksize * Cin, p->Kp = rup(p->K, 32), p->Np = rup(Cout, 128);
  const bool wino = ksize == 3 && Cin % 16 == 0 && c->tn.use_wino;
  const size_t panel = (size_t)p->Np * p->Kp, total = panel + p->Np + (wino ? wino_u_floats(Cout, Cin) : 0);
  hipError_t e = hipMalloc((void**)&p->wp, total * sizeof(float));
  if (e != hipSuccess) {
    delete p;
    return fail(c, MGU_ERR_NOMEM, "hipMalloc(%zu) failed: %s", total * sizeof(float), hipGetErrorString(e));
  }
  p->shift = p->wp + panel;
  e = hipMemsetAsync(p->wp, 0, (panel + p->Np) * sizeof(float), s);
  if (e == hipSuccess) e = launch_pack_conv_w((const float*)w_dev, p->wp, 0, Cout, Cin, Cin, ksize, p->Kp, s);
  if (e == hipSuccess && wino) {
    p->wu = p->shift + p->Np;
    e = launch_pack_wino_w((const float*)w_dev, p->wu, Cout, Cin, Cin, 0, c->tn.wino_prec, s);
  }
  if (e != hipSuccess) {
    (void)hipFree(p->wp);
    delete p;
    return fail(c, MGU_ERR_HIP, "conv2d_prepare: %s", hipGetErrorString(e));
  }
  *out = p;
  return MGU_OK;
}

void mgu_conv2d_release(mgu_ctx* c, mgu_conv_weights* p) {
  if (!p) return;
  if (c) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();   // launches that read the panels may still be in flight
  if (p->wp) (void)hipFree(p->wp);
  delete p;
}

int mgu_conv2d_prepared_nhwc(mgu_ctx* c, const mgu_conv_weights* p, const void* in_dev, int B, int H, int W, const void* bias_dev,
                             const void* scale_dev, const void* shift_dev, int relu, void* out_dev, int ld_out, int c_off,
                             void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!p || !in_dev || !out_dev || B < 1 || H < 1 || W < 1) return fail(c, MGU_ERR_INVALID, "bad conv2d args");
  if (ld_out < c_off + p->Cout) return fail(c, MGU_ERR_INVALID, "ld_out %d < c_off %d + Cout %d", ld_out, c_off, p->Cout);
  if ((int64_t)B * H * W >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = (const float*)in_dev, d.w = p->wp, d.wu = p->wu, d.out = (float*)out_dev;
  d.M = B * H * W, d.H = H, d.W = W, d.Cp = p->Cin, d.ldin = p->Cin, d.KS = p->ksize, d.K = p->K, d.Kp = p->Kp;
  d.N = p->Cout, d.ldout = ld_out, d.coff = c_off, d.relu = relu;
  if (scale_dev && shift_dev) {  // y = scale*(conv) + shift, bias folded by the caller into shift
    d.scale = (const float*)scale_dev;
    d.shift = (const float*)shift_dev;
  } else if (bias_dev) {
    HIPCHK(c, launch_bias_tile((const float*)bias_dev, p->shift, p->Cout, 1, s));
    d.shift = p->shift;
  }
  ProfScope ps(c, s);
  HIPCHK(c, launch_igemm_f32(d, s));
  return MGU_OK;
}

// One-shot form: packs the weight on EVERY call (parity tests, weights that change between calls); steady-state callers
// use mgu_conv2d_prepare + mgu_conv2d_prepared_nhwc.
int mgu_conv2d_nhwc(mgu_ctx* c, const void* in_dev, int B, int H, int W, int Cin, const void* w_dev, const void* bias_dev,
                    const void* scale_dev, const void* shift_dev, int Cout, int ksize, int relu, void* out_dev,
                    int ld_out, int c_off, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!in_dev || !w_dev || !out_dev || B < 1 || H < 1 || W < 1 || Cout < 1 || (ksize != 1 && ksize != 3))
    return fail(c, MGU_ERR_INVALID, "bad conv2d args (ksize must be 1 or 3)");
  if (Cin < 4 || (Cin & 3)) return fail(c, MGU_ERR_INVALID, "conv2d needs Cin %% 4 == 0 (got %d)", Cin);
  if (ld_out < c_off + Cout) return fail(c, MGU_ERR_INVALID, "ld_out %d < c_off %d + Cout %d", ld_out, c_off, Cout);
  if ((int64_t)B * H * W >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  Layer L;
  L.Cin = L.Cp = Cin, L.Cout = Cout, L.KS = ksize;
  L.K = ksize * ksize * Cin, L.Kp = rup(L.K, 32), L.N = Cout, L.Np = rup(Cout, 128);
  float *sc, *sh;
  int rc = block_scratch(c, L.Np, L.Kp, &L.wp, &sc, &sh, s);
  if (rc) return rc;
  HIPCHK(c, launch_pack_conv_w((const float*)w_dev, L.wp, 0, Cout, Cin, Cin, ksize, L.Kp, s));
  if (ksize == 3 && Cin % 16 == 0 && c->tn.use_wino) {   // same routing as the model's layers: Winograd F(2x2,3x3)
    if ((rc = ensure(c, &c->wuws, &c->wuws_bytes, wino_u_floats(Cout, Cin) * sizeof(float)))) return rc;
    L.wu = (float*)c->wuws;
    HIPCHK(c, launch_pack_wino_w((const float*)w_dev, L.wu, Cout, Cin, Cin, 0, c->tn.wino_prec, s));
  }
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = (const float*)in_dev, d.w = L.wp, d.wu = L.wu, d.out = (float*)out_dev;
  d.M = B * H * W, d.H = H, d.W = W, d.Cp = Cin, d.ldin = Cin, d.KS = ksize, d.K = L.K, d.Kp = L.Kp;
  d.N = Cout, d.ldout = ld_out, d.coff = c_off, d.relu = relu;
  if (scale_dev && shift_dev) {  // y = scale*(conv) + shift, bias folded by the caller into shift
    d.scale = (const float*)scale_dev;
    d.shift = (const float*)shift_dev;
  } else if (bias_dev) {
    HIPCHK(c, launch_bias_tile((const float*)bias_dev, sh, Cout, 1, s));
    d.shift = sh;
  }
  ProfScope ps(c, s);
  HIPCHK(c, launch_igemm_f32(d, s));
  return MGU_OK;
}

int mgu_conv_transpose2x2_nhwc(mgu_ctx* c, const void* in_dev, int B, int H, int W, int Cin, const void* w_dev,
                               const void* bias_dev, int Cout, void* out_dev, int ld_out, int c_off, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!in_dev || !w_dev || !out_dev || B < 1 || H < 1 || W < 1 || Cout < 1) return fail(c, MGU_ERR_INVALID, "bad convT args");
  if (Cin < 4 || (Cin & 3)) return fail(c, MGU_ERR_INVALID, "convT needs Cin %% 4 == 0 (got %d)", Cin);
  if (ld_out < c_off + Cout) return fail(c, MGU_ERR_INVALID, "ld_out %d < c_off %d + Cout %d", ld_out, c_off, Cout);
  if ((int64_t)B * H * W * 4 >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "4*B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int Kp = rup(Cin, 32), N = 4 * Cout, Np = rup(N, 128);
  float *wp, *sc, *sh;
  // Two packed forms, each in a region of its own: the direct [4 Cout][Cin] panel the tile kernel reads (always built), and -- when
  // the layer shape is eligible and the context's switches allow it (MGU_NO_CONVT_FRAG, MGU_WINO_PREC) -- the fragment-order
  // three-piece weights of convt2x2_x3_kernel.  Which kernel runs is decided by the dispatcher (convt_x3_applicable) on the
  // COMPLETE descriptor; a launch it rejects falls back to the tile kernel, which then finds a real panel in d.w.
  const bool x3_shape = Cin % 32 == 0 && Cout % 32 == 0 && c->tn.wino_prec != 0 && c->tn.convt_frag;
  const size_t panel = (size_t)Np * Kp + 2 * (size_t)Np;
  int rc = ensure(c, &c->gws, &c->gws_bytes, (panel + (x3_shape ? convt_x3_floats(Cin, Cout) : 0)) * sizeof(float));
  if (rc) return rc;
  wp = (float*)c->gws, sc = wp + (size_t)Np * Kp, sh = sc + Np;
  (void)sc;
  HIPCHK(c, hipMemsetAsync(c->gws, 0, panel * sizeof(float), s));
  HIPCHK(c, launch_pack_convt_w((const float*)w_dev, wp, 0, Cin, Cout, Kp, s));
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = (const float*)in_dev, d.w = wp, d.out = (float*)out_dev;
  d.M = B * H * W, d.H = H, d.W = W, d.Cp = Cin, d.ldin = Cin, d.KS = 1, d.K = Cin, d.Kp = Kp;
  d.N = N, d.ldout = ld_out, d.coff = c_off, d.out_mode = 1, d.ct_cout = Cout, d.Hout = 2 * H, d.Wout = 2 * W;
  if (bias_dev) {
    HIPCHK(c, launch_bias_tile((const float*)bias_dev, sh, Cout, 4, s));
    d.shift = sh;
  }
  if (x3_shape) {
    float* wx = wp + panel;
    d.wu = wx;
    if (convt_x3_applicable(d)) HIPCHK(c, launch_pack_convt_x3((const float*)w_dev, wx, Cin, Cout, s));
    else d.wu = nullptr;
  }
  ProfScope ps(c, s);
  HIPCHK(c, launch_igemm_f32(d, s));
  return MGU_OK;
}

int mgu_maxpool2x2_nhwc(mgu_ctx* c, const void* in_dev, int ld_in, int B, int H, int W, int Cc, void* out_dev,
                        void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!in_dev || !out_dev || B < 1 || H < 2 || W < 2 || Cc < 4 || (Cc & 3) || ld_in < Cc || (ld_in & 3))
    return fail(c, MGU_ERR_INVALID, "bad maxpool args (C and ld_in must be multiples of 4, H,W >= 2)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_maxpool2(in_dev, ld_in, out_dev, 0, B, H, W, Cc, (hipStream_t)hip_stream));
  return MGU_OK;
}

int mgu_argmax_classes(mgu_ctx* c, const void* logits_dev, int64_t npix, int num_classes, int64_t* pred_dev,
                       void* hip_stream) {
  if (!c || !logits_dev || !pred_dev || num_classes < 1 || npix < 0) return fail(c, MGU_ERR_INVALID, "bad argmax args");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_argmax((const float*)logits_dev, npix, num_classes, pred_dev, (hipStream_t)hip_stream));
  return MGU_OK;
}

int mgu_patch_mean(mgu_ctx* c, const void* feat_dev, int feat_dtype, int B, int H, int W, int C, int patch, void* out_dev,
                   void* hip_stream) {
  if (!c || !feat_dev || !out_dev || B < 1 || H < 1 || W < 1 || patch < 1)
    return fail(c, MGU_ERR_INVALID, "bad patch_mean args");
  if (feat_dtype != MGU_DTYPE_F32 && feat_dtype != MGU_DTYPE_BF16) return fail(c, MGU_ERR_INVALID, "unknown dtype %d", feat_dtype);
  const int vec = feat_dtype == MGU_DTYPE_BF16 ? 8 : 4;
  if ((C % vec) || C < vec || C > 256)
    return fail(c, MGU_ERR_INVALID, "patch_mean needs C %% %d == 0 and C <= 256 (got %d)", vec, C);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_patch_mean(feat_dev, feat_dtype, (float*)out_dev, B, H, W, C, patch, (hipStream_t)hip_stream));
  return MGU_OK;
}

double mgu_unet_flops(mgu_ctx* c, int B, int H, int W) {
  if (!c || !c->configured) return -1.0;
  std::vector<int> hs, wsz;
  level_dims(H, W, c->depth, hs, wsz);
  double fl = 0;
  int li = 0;
  for (int i = 0; i <= c->depth; ++i)
    for (int j = 0; j < 2; ++j, ++li) {
      const Layer& L = c->layers[li];
      fl += 2.0 * hs[i] * wsz[i] * 9.0 * L.Cin * L.Cout;
    }
  for (int b = 0; b < c->depth; ++b) {
    const int i = c->depth - 1 - b;
    const Layer& U = c->layers[li++];
    fl += 2.0 * hs[i + 1] * wsz[i + 1] * (double)U.Cin * U.Cout * 4.0;
    for (int j = 0; j < 2; ++j, ++li) {
      const Layer& L = c->layers[li];
      fl += 2.0 * hs[i] * wsz[i] * 9.0 * L.Cin * L.Cout;
    }
  }
  fl += 2.0 * H * W * (double)c->layers[li].Cin * c->ncls;
  return fl * B;
}

double mgu_unet_mfma_flops(mgu_ctx* c, int B, int H, int W) {
  if (!c || !c->configured) return -1.0;
  std::vector<int> hs, wsz;
  level_dims(H, W, c->depth, hs, wsz);
  auto conv = [&](const Layer& L, int h, int w) {
    if (L.wino && L.wu && c->tn.use_wino) return 2.0 * ((h + 1) / 2) * ((w + 1) / 2) * 16.0 * L.Cp * L.Cout;   // per 2x2 tile: 16 products
    return 2.0 * h * w * 9.0 * L.Cin * L.Cout;
  };
  double fl = 0;
  int li = 0;
  for (int i = 0; i <= c->depth; ++i)
    for (int j = 0; j < 2; ++j, ++li) fl += conv(c->layers[li], hs[i], wsz[i]);
  for (int b = 0; b < c->depth; ++b) {
    const int i = c->depth - 1 - b;
    const Layer& U = c->layers[li++];
    fl += 2.0 * hs[i + 1] * wsz[i + 1] * (double)U.Cin * U.Cout * 4.0;
    for (int j = 0; j < 2; ++j, ++li) fl += conv(c->layers[li], hs[i], wsz[i]);
  }
  fl += 2.0 * H * W * (double)c->layers[li].Cin * c->ncls;
  return fl * B;
}

int mgu_profile_enable(mgu_ctx* c, int on) {
  if (!c) return MGU_ERR_INVALID;
  c->prof = on != 0;
  c->ev_used = 0;
  return MGU_OK;
}

int mgu_profile_read_kernels(mgu_ctx* c, mgu_kernel_stat* out, int cap, int* n_out) {
  if (!c || !n_out || (cap > 0 && !out)) return MGU_ERR_INVALID;
  int n = 0;
  for (int i = 0; i < c->ev_used; ++i) {
    float ms = 0;
    HIPCHK(c, hipEventSynchronize(c->ev[2 * i + 1]));
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
    const mgu_ctx::ProfRec& r = c->prec[i];
    int k = 0;
    while (k < n && strcmp(out[k].name, r.name) != 0) ++k;
    if (k == n) {
      if (n == cap) continue;   // table full: the remaining families are dropped (cap >= 32 holds every family of a step)
      out[n].name = r.name, out[n].ms = 0, out[n].flops_alg = 0, out[n].flops_mfma = 0, out[n].launches = 0, out[n].pipe = r.pipe;
      ++n;
    }
    out[k].ms += ms, out[k].flops_alg += r.alg, out[k].flops_mfma += r.mfma, out[k].launches += 1;
  }
  *n_out = n;
  c->ev_used = 0;   // a read consumes the records: with profiling left on, the next window starts from an empty table
  return MGU_OK;
}

int mgu_profile_read(mgu_ctx* c, double* conv_ms, int* conv_launches, double* total_ms) {
  if (!c) return MGU_ERR_INVALID;
  if (c->ev_used == 0 || !c->ev_total[1]) return fail(c, MGU_ERR_STATE, "no profiled forward to read");
  HIPCHK(c, hipEventSynchronize(c->ev_total[1]));
  double sum = 0;
  for (int i = 0; i < c->ev_used; ++i) {
    float ms = 0;
    HIPCHK(c, hipEventSynchronize(c->ev[2 * i + 1]));
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
    sum += ms;
  }
  float tot = 0;
  HIPCHK(c, hipEventElapsedTime(&tot, c->ev_total[0], c->ev_total[1]));
  if (conv_ms) *conv_ms = sum;
  if (conv_launches) *conv_launches = c->ev_used;
  if (total_ms) *total_ms = tot;
  return MGU_OK;
}

}  // extern "C"
