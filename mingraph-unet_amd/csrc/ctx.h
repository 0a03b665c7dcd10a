// Internal: context and helpers shared by mgunet_api.hip and mgunet_train.hip (not part of the ABI).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mgunet.h"
#include "common.h"

struct Layer {
  std::string prefix;   // state_dict prefix, e.g. "encoder.encoder_blocks.0."
  std::string conv;     // "conv1" | "conv2" | "upsample" | "final_conv"
  std::string bn;       // "bn1" | "bn2" | ""
  int Cin = 0, Cp = 0, Cout = 0, KS = 3;
  bool convt = false;
  int K = 0, Kp = 0, N = 0, Np = 0;
  float *wp = nullptr, *scale = nullptr, *shift = nullptr;
  bool first = false;    // fp32 first convolution (<= 4 input channels): VALU kernel, weights kept in wf
  bool ctb = false;      // bf16-storage ConvTranspose with Cin % 64 == 0, Cout % 32 == 0: bf16 fragment weights kept in wu (convt_bf16.hip)
  bool ctx3 = false;     // fp32 ConvTranspose with Cin % 32 == 0, Cout % 32 == 0: three-piece fragment weights kept in wu (convt_x3.hip)
  bool wino = false;     // fp32 3x3 layer with Cp % 16 == 0: Winograd-transformed weights kept in wu
  mutable bool wp_dirty = false;   // direct panel wp not yet rebuilt from w_src (packed on first use)
  float* wu = nullptr;
  float* wug = nullptr;          // Winograd weights of the layer's DATA-GRADIENT convolution (training; same launch as wu)
  mutable bool wug_valid = false;   // wug holds the current weights
  float* wf = nullptr;   // first conv (Cp == 4): [9][4][Cout] weights for conv3x3_first_kernel
  float* wfm = nullptr;  // the same layer's three-piece fragment weights for conv3x3_first_mfma_kernel (Cin <= 3, Cout == 32)
  float* wxg = nullptr;  // data-gradient weights kept across the step (training; packed with everything else by the weight refresh):
                         // three-piece fragments of a ConvTranspose (convt_x3.hip) or the direct panel of the final 1x1 conv
  mutable bool wxg_valid = false;
  // caller-owned parameter tensors recorded by load_weights (used by the training path)
  const float *w_src = nullptr, *b_src = nullptr, *gamma = nullptr, *beta = nullptr;
  float *run_mean = nullptr, *run_var = nullptr;
  // offsets (elements) into the flat parameter / gradient vector, named_parameters() order
  int64_t off_w = 0, off_b = 0, off_gamma = 0, off_beta = 0;
  // batch statistics of the last training forward (library scratch)
  float *mean = nullptr, *invstd = nullptr, *tscale = nullptr, *tshift = nullptr;
  // tensors of the last training forward
  const float* t_in = nullptr;  // conv input
  int t_ldin = 0;
  float* t_z = nullptr;         // raw conv output (dense, pitch Cout)
  float* t_y = nullptr;         // relu(bn(z)) with pitch t_ldy
  int t_ldy = 0;
  int t_B = 0, t_H = 0, t_W = 0;  // grid the conv ran on (input grid for convT)
};

struct mgu_ctx {
  int device = 0;
  mgu::Tuning tn;           // kernel-selection switches of THIS context (MGU_* environment at mgu_create)
  std::string err;
  bool configured = false, loaded = false;
  std::vector<mgu::WinoPackBatch> pack_host;   // the Winograd pack tables as last uploaded (repack_weights)
  mgu::WinoPackBatch* pack_dev = nullptr;
  int pack_dev_cap = 0;
  bool want_train = false;  // a training forward has run on this context: weight refreshes also build the data-gradient forms
  bool fold_dirty = false;  // BN running stats / affine changed since the eval scale/shift were folded
  int* err_word = nullptr;  // host-mapped word a kernel sets when it meets invalid DATA (e.g. a label out of range): read by
                            // mgu_sync_check and, without a sync, at the entry of the next training call
  void* redws = nullptr;    // per-channel reduction slots (self-cleaning: zero between launches)
  size_t redws_bytes = 0;
  int last_stat_rows = 0;   // accumulator rows the last statistics-fused conv launch wrote (run_layer)
  void* pm_out = nullptr;   // one-shot request (mgu_unet_request_patch_mean): patch means of decoder feature 0
  int pm_patch = 0;
  void* wuws = nullptr;     // Winograd weight scratch of the mgu_conv2d_nhwc building block
  size_t wuws_bytes = 0;
  unsigned long long* gmaxbuf = nullptr;   // GAT: [64 slots][gmax_cap] per-(graph, head) max accumulators, (generation, value) words
  int gmax_cap = 0;
  unsigned gmax_gen = 0;                   // generation of the last layer call (gat_common.h)
  struct mgu_gat_weights* gat_tmp = nullptr;   // weights prepared by the one-shot mgu_gat_layer_forward
  void* lossws = nullptr;   // partial records of the auxiliary-loss reductions (losses.hip)
  size_t lossws_bytes = 0;
  void* imgws = nullptr;    // resampling coefficient tables / histograms of the input pipeline (imageops.hip)
  size_t imgws_bytes = 0;
  void* ncws = nullptr;     // normalized-cut accumulators (mgu_ncut_forward)
  size_t ncws_bytes = 0;
  int in_ch = 0, ncls = 0, feat = 0, depth = 0, dtype = 0, Cp0 = 0;
  std::vector<Layer> layers;  // enc[i].conv1, enc[i].conv2 ..., bott.conv1, bott.conv2, dec[b].up, dec[b].conv1, dec[b].conv2 ..., final
  int64_t nparams = 0;
  float* arena = nullptr;
  size_t arena_floats = 0;
  void* ws = nullptr;   // eval scratch
  size_t ws_bytes = 0;
  void* gbws = nullptr;     // GAT backward scratch (gat_bwd.hip)
  size_t gbws_bytes = 0;
  void* gbpanel = nullptr;  // GAT backward: the linear layer's packed panel
  size_t gbpanel_bytes = 0;
  void* gws = nullptr;  // GAT / building-block scratch
  size_t gws_bytes = 0;
  void* tws = nullptr;  // training scratch (saved activations + backward temporaries)
  size_t tws_bytes = 0;
  // last training forward
  bool have_train_fwd = false;
  int tB = 0, tH = 0, tW = 0;
  std::vector<float*> t_cat, t_feat, t_pooled;
  float* t_logits = nullptr;
  void gat_destroy(mgu_ctx* c);   // gat_api.hip
int gmax_buffer(mgu_ctx* c, int need, unsigned long long** buf, unsigned* gen);
int gat_linear_st(mgu_ctx* c, const float* X, int N, int Fin, const float* W, const float* a, int heads, int Fh, float* wh, float* st,
                  hipStream_t s);

// gradient exchange (comm.hip): RCCL communicator owned by this context, its stream and a small pool of ordering events
  void* comm = nullptr;     // ncclComm_t
  int comm_world = 1, comm_rank = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev[16] = {};
  int comm_ev_next = 0;
  // profiling: HIP event pairs on the launch stream around the launches recorded since mgu_profile_enable(ctx, 1)
  bool prof = false;
  std::vector<hipEvent_t> ev;  // pairs
  struct ProfRec {
    const char* name;     // kernel (family) name, static storage
    double alg, mfma;     // algorithmic FLOPs (2*MAC of the operator) / FLOPs issued on the matrix pipe
    int pipe;             // matrix pipe: 0 = fp32 MFMA, 1 = bf16 MFMA, -1 = none (VALU / bandwidth kernels)
  };
  std::vector<ProfRec> prec;
  int ev_used = 0;
  hipEvent_t ev_total[2] = {nullptr, nullptr};
};

namespace mgud {

inline int rup(int v, int m) { return (v + m - 1) / m * m; }
extern std::string g_create_err;

inline int fail(mgu_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return code;
}

#define HIPCHK(c, call)                                                                                       \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return mgud::fail(c, MGU_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
  } while (0)

inline int ensure(mgu_ctx* c, void** p, size_t* have, size_t need) {
  if (*have >= need) return MGU_OK;
  if (*p) {
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  hipError_t e = hipMalloc(p, need);
  if (e != hipSuccess) return fail(c, MGU_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
  *have = need;
  return MGU_OK;
}

// device pointer of the context's host-mapped data-error word (allocated on first use).  Bits: 1 = a label outside [0, num_classes)
// reached a loss kernel; 2 = the GAT backward found no edge equal to the forward's graph-wide maximum (gat_bwd.hip)
inline int err_word_dev(mgu_ctx* c, int** dev) {
  if (!c->err_word) {
    HIPCHK(c, hipHostMalloc((void**)&c->err_word, sizeof(int), hipHostMallocMapped));
    *c->err_word = 0;
  }
  HIPCHK(c, hipHostGetDevicePointer((void**)dev, c->err_word, 0));
  return MGU_OK;
}
// message of a pending data error (word value w)
inline const char* err_word_message(int w) {
  return (w & 2) ? "the GAT backward found no edge whose score equals the forward's graph-wide maximum (mgu_gat_layer_backward): its "
                   "max term was not applied"
                 : "a label outside [0, num_classes) (and != ignore_index -100) reached a loss kernel of this context (mgu_cross_entropy / "
                   "mgu_dice_loss; F.one_hot / CrossEntropyLoss raise on it)";
}

struct ProfScope {  // records an event pair around one launch when profiling is on
  mgu_ctx* c;
  hipStream_t s;
  int idx = -1;
  ProfScope(mgu_ctx* c_, hipStream_t s_, const char* name = "conv/GEMM", double alg = 0, double mfma = 0, int pipe = -1) : c(c_), s(s_) {
    if (!c->prof) return;
    if ((size_t)c->ev_used >= c->prec.size()) c->prec.resize(c->ev_used + 1);
    c->prec[c->ev_used] = {name, alg, mfma, pipe};
    if ((size_t)(2 * c->ev_used + 2) > c->ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->ev.push_back(a);
      c->ev.push_back(b);
    }
    idx = c->ev_used++;
    (void)hipEventRecord(c->ev[2 * idx], s);
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(c->ev[2 * idx + 1], s);
  }
};

inline void level_dims(int H, int W, int depth, std::vector<int>& hs, std::vector<int>& wsz) {
  hs.assign(depth + 1, 0);
  wsz.assign(depth + 1, 0);
  hs[0] = H;
  wsz[0] = W;
  for (int i = 1; i <= depth; ++i) {
    hs[i] = hs[i - 1] / 2;  // MaxPool2d(2,2) floor mode, unet_encoder.py:48
    wsz[i] = wsz[i - 1] / 2;
  }
}

// one fused conv / convT / 1x1 launch described by a Layer (scale/shift chosen by the caller)
int run_layer(mgu_ctx* c, const Layer& L, const void* in, int ldin, int B, int H, int W, void* out, int ldout, int coff,
              int relu, const float* scale, const float* shift, int Hout, int Wout, hipStream_t s,
              void* pool = nullptr, int ldpool = 0, bool* pool_fused = nullptr,   // optional fused MaxPool2d(2) output
              double* stat_slots = nullptr, bool* stat_fused = nullptr);          // optional fused BatchNorm batch statistics

void gat_destroy(mgu_ctx* c);   // gat_api.hip
int gmax_buffer(mgu_ctx* c, int need, unsigned long long** buf, unsigned* gen);
int gat_linear_st(mgu_ctx* c, const float* X, int N, int Fin, const float* W, const float* a, int heads, int Fh, float* wh, float* st,
                  hipStream_t s);

// gradient exchange (comm.hip)
int comm_bucket(mgu_ctx* c, float* flat, int64_t lo, int64_t hi, hipStream_t s);
int comm_join(mgu_ctx* c, hipStream_t s);

int repack_weights(mgu_ctx* c, hipStream_t s);   // mgunet_api.hip: every packed weight form from the recorded parameter tensors
// training path (mgunet_train.hip)
size_t train_ws_bytes(const mgu_ctx* c, int B, int H, int W);
int unet_forward_train(mgu_ctx* c, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int64_t xs_w, int B, int H,
                       int W, float* logits, void* const* cat_dev, void* const* feat_dev, hipStream_t s);

}  // namespace mgud
