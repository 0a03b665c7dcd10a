/*
 * mgunet.h -- C-ABI of libmgunet.so: the MI355X (gfx950) MinGraph-UNet segmentation hot path.
 *
 * The reference (agent-charon/MinGraph-UNet) has no FFI, plugin or operator registry: its hot
 * path sits behind three Python classes.  Each entry point below names the reference interface
 * it replaces (paths relative to MinGraph-UNet/).  A maintainer binds these with ctypes -- see
 * INTEGRATION.md for the stub that swaps them in under the reference's own nn.Modules.
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is a HIP device pointer owned by the caller;
 *   - activations are NHWC ("channels_last") fp32 in device memory; logical shapes stay NCHW;
 *   - all work is enqueued on `hip_stream` (a hipStream_t passed as void*); no hidden syncs
 *     except inside mgu_*_reserve / the first call that has to grow the library-owned workspace;
 *   - every function returns MGU_OK (0) or a negative error code; mgu_last_error(ctx) gives text;
 *   - a ctx is bound to one device and is not thread-safe.
 */
#ifndef MGUNET_H
#define MGUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGU_OK 0
#define MGU_ERR_INVALID -1   /* bad argument / unsupported shape (shim raises ValueError)   */
#define MGU_ERR_HIP -2       /* a HIP runtime call failed (shim raises RuntimeError)          */
#define MGU_ERR_STATE -3     /* call order violated, e.g. forward before load_weights         */
#define MGU_ERR_NOMEM -4

#define MGU_DTYPE_F32 0
#define MGU_DTYPE_BF16 1     /* bf16 storage + fp32 accumulate, inference only (BASELINE config 3) */

typedef struct mgu_ctx mgu_ctx;

/* One named parameter/buffer of a state_dict(); `ptr` is a device pointer to contiguous fp32
 * (int64 for num_batches_tracked, which is ignored).  Names are the reference's state_dict keys
 * (SURVEY.md 8b): "encoder.encoder_blocks.0.conv1.weight", "gat_layers.0.heads.2.a.weight", ... */
typedef struct {
  const char* name;
  const void* ptr;
  int64_t numel;
} mgu_tensor_desc;

/* ---- context ------------------------------------------------------------------------------ */
int mgu_create(int device_id, mgu_ctx** out);
void mgu_destroy(mgu_ctx* ctx);
const char* mgu_last_error(mgu_ctx* ctx); /* ctx may be NULL: returns the last create() error */
const char* mgu_version(void);

/* ---- U-Net: replaces model/unet/unet_model.py:6-36 (UNet.__init__/forward) ------------------- */
/* = UNet(in_channels, num_classes, init_features, depth) (unet_model.py:7).  init_features must be
 * a multiple of 4 (NHWC 16-byte lanes).  dtype MGU_DTYPE_BF16 (init_features % 8 == 0, <= 4 classes): every
 * activation the forward writes (cat_dev, feat_dev) and the packed weights are bf16, accumulation and the
 * bias/BatchNorm epilogue stay fp32, x_dev and logits_dev stay fp32; eval mode only. */
int mgu_unet_configure(mgu_ctx* ctx, int in_channels, int num_classes, int init_features, int depth, int dtype);
/* Number of fp32 elements of all trainable parameters, in state_dict order (7 766 018 for (3,2,32,4)). */
int64_t mgu_unet_param_count(mgu_ctx* ctx);
/* = load_state_dict: repacks OIHW conv weights to the kernels' [Cout][tap][Cin] panels, and for
 * eval folds bias + BatchNorm running stats (unet_encoder.py:12-13, eps 1e-5) into a per-channel
 * scale/shift applied in the conv epilogue.  Must be re-called after the parameters change. */
int mgu_unet_load_weights(mgu_ctx* ctx, const mgu_tensor_desc* named, int n, void* hip_stream);
/* The parameter tensors recorded by the last mgu_unet_load_weights were modified IN PLACE (same addresses: an optimizer step
 * through the flat parameter buffer, scripts/train_segmentation.py:134): rebuild every packed weight form from them -- one launch
 * for all Winograd sets (forward and, once the context has trained, data-gradient), the eval BatchNorm fold stays lazy.  The
 * train step calls this instead of going through the named state_dict again. */
int mgu_unet_refresh_weights(mgu_ctx* ctx, void* hip_stream);

/* Bytes of library-owned scratch a forward of this shape uses (allocated on first use).  training = 1: the train-mode
 * forward + backward scratch (every layer's pre-activation and activation, gradient temporaries, weight-gradient partial
 * panels: about 1.9 GB at 4 x 3x512x512), a separate allocation from the eval scratch. */
int mgu_unet_workspace_bytes(mgu_ctx* ctx, int B, int H, int W, int training, size_t* out);
/* Pre-allocate that scratch (synchronous); forward does it lazily otherwise. */
int mgu_unet_reserve(mgu_ctx* ctx, int B, int H, int W, int training);
/* = logits, skips, dec_feats = UNet.forward(x)  (unet_model.py:34-36); training=0: eval mode (BatchNorm folded),
 * training=1: train mode (see "training step" below).
 *   x_dev      : input, element (n,c,y,x) at x_dev[n*xs_n + c*xs_c + y*xs_h + x*xs_w] (fp32; any layout)
 *   logits_dev : (B,H,W,num_classes) NHWC
 *   cat_dev[i] : i = 0..depth-1 shallow->deep, NHWC buffer (B,H_i,W_i,2*C_i), C_i = init_features<<i,
 *                H_i = H>>i.  Channels [0,C_i) receive skip connection i (unet_encoder.py:69); channels
 *                [C_i,2*C_i) receive the up-sampled decoder input (unet_decoder.py:36-53), i.e. the
 *                buffer IS torch.cat([skip, up], 1) and no concat kernel ever runs.
 *   feat_dev[i]: decoder feature i shallow->deep, NHWC dense (B,H_i,W_i,C_i) (unet_decoder.py:141,149)
 * All outputs stay valid after the call (caller-owned). */
int mgu_unet_forward(mgu_ctx* ctx, const void* x_dev, int B, int H, int W,
                     int64_t xs_n, int64_t xs_c, int64_t xs_h, int64_t xs_w,
                     void* logits_dev, void* const* cat_dev, void* const* feat_dev,
                     int training, void* hip_stream);
/* Building blocks (also used by the kernel-level parity tests): one fused Conv2d(k=1|3, pad=k/2, bias)
 * [+ per-channel scale/shift] [+ ReLU] on an NHWC fp32 tensor, = nn.Conv2d.forward as used at
 * unet_encoder.py:7-8,16-24 and unet_decoder.py:117; weights in the reference's OIHW layout (device).
 * scale_dev/shift_dev may be NULL (then y = conv + bias).  in: (B,H,W,Cin) with Cin % 4 == 0; out: pixel
 * pitch ld_out >= c_off + Cout floats (lets the result land in a channel slice of a wider buffer). */
int mgu_conv2d_nhwc(mgu_ctx* ctx, const void* in_dev, int B, int H, int W, int Cin, const void* w_oihw_dev,
                    const void* bias_dev, const void* scale_dev, const void* shift_dev, int Cout, int ksize,
                    int relu, void* out_dev, int ld_out, int c_off, void* hip_stream);
/* Steady-state form of mgu_conv2d_nhwc: mgu_conv2d_nhwc repacks the OIHW weight into the kernels' panels (and the Winograd
 * transform) on EVERY call; a caller whose weights are constant between calls (an eval-mode nn.Conv2d, e.g. the two
 * convolutions of DetectionHead, detection_head.py:33,36) packs them once.  The handle is owned by the library and stays
 * valid until mgu_conv2d_release; re-prepare after the weight tensor changes.  bias/scale/shift as for mgu_conv2d_nhwc. */
typedef struct mgu_conv_weights mgu_conv_weights;
int mgu_conv2d_prepare(mgu_ctx* ctx, const void* w_oihw_dev, int Cout, int Cin, int ksize, mgu_conv_weights** out, void* hip_stream);
void mgu_conv2d_release(mgu_ctx* ctx, mgu_conv_weights* w);
int mgu_conv2d_prepared_nhwc(mgu_ctx* ctx, const mgu_conv_weights* w, const void* in_dev, int B, int H, int W, const void* bias_dev,
                             const void* scale_dev, const void* shift_dev, int relu, void* out_dev, int ld_out, int c_off,
                             void* hip_stream);
/* ConvTranspose2d(Cin, Cout, kernel_size=2, stride=2) + bias (unet_decoder.py:25,36); weight (Cin,Cout,2,2).
 * in (B,H,W,Cin) NHWC -> out (B,2H,2W,*) NHWC with pixel pitch ld_out, channels [c_off, c_off+Cout). */
int mgu_conv_transpose2x2_nhwc(mgu_ctx* ctx, const void* in_dev, int B, int H, int W, int Cin, const void* w_iohw_dev,
                               const void* bias_dev, int Cout, void* out_dev, int ld_out, int c_off, void* hip_stream);
/* MaxPool2d(2,2) floor mode (unet_encoder.py:48) on NHWC, input pixel pitch ld_in >= C. */
int mgu_maxpool2x2_nhwc(mgu_ctx* ctx, const void* in_dev, int ld_in, int B, int H, int W, int C, void* out_dev,
                        void* hip_stream);
/* argmax over classes of NHWC logits -> int64 (B,H,W): experiments/segmentation_performance.py:141 */
int mgu_argmax_classes(mgu_ctx* ctx, const void* logits_dev, int64_t npix, int num_classes,
                       int64_t* pred_dev, void* hip_stream);

/* ---- training step: replaces the autograd graph of scripts/train_segmentation.py:121-134 ---------- */
/* mgu_unet_forward(training=1) is the train-mode forward: BatchNorm uses batch statistics (biased
 * variance), updates running_mean/var IN PLACE in the tensors given to mgu_unet_load_weights (momentum
 * 0.1, unbiased variance; unet_encoder.py:12-13) and keeps the activations backward needs in library
 * scratch.  The forward's caller-owned outputs must stay alive until mgu_unet_backward returns.
 *
 * Offset (elements) of a parameter in the flat parameter/gradient vector; the order is the reference's
 * named_parameters() order.  name = state_dict key of a weight/bias; -1 if unknown. */
int64_t mgu_unet_param_offset(mgu_ctx* ctx, const char* name);
/* nn.CrossEntropyLoss() (mean reduction, ignore_index = -100: torch's defaults) forward + gradient
 * (train_segmentation.py:91,127): logits_dev (npix, C) NHWC fp32, labels_dev int64 (npix).
 * Writes *loss_dev = mean over the counted pixels of -log softmax(l_i)[y_i] and dlogits_dev (npix, 4*ceil(C/4)) =
 * grad_scale * npix/count * (softmax - onehot), pad columns zero: grad_scale = 1/npix reproduces loss.backward() of the
 * mean loss.  A pixel labelled -100 is not counted and gets a zero gradient (torch semantics).  Any other label outside
 * [0, C) is invalid DATA, which the host cannot see without a synchronisation: the label is never used as an index, the
 * loss comes out NaN, and the NEXT mgu_cross_entropy / mgu_unet_backward / mgu_sync_check on this ctx that runs after the
 * kernel has executed returns MGU_ERR_INVALID (torch raises IndexError / device-asserts at the same point). */
int mgu_cross_entropy(mgu_ctx* ctx, const void* logits_dev, const int64_t* labels_dev, int64_t npix, int num_classes,
                      float grad_scale, void* dlogits_dev, float* loss_dev, void* hip_stream);
/* Synchronise hip_stream and report invalid data met by kernels of this ctx since the last check (MGU_ERR_INVALID). */
int mgu_sync_check(mgu_ctx* ctx, void* hip_stream);
/* loss.backward() (train_segmentation.py:133) for the last mgu_unet_forward(training=1): dlogits_dev as
 * produced by mgu_cross_entropy; every element of flat_grad_dev (mgu_unet_param_count floats) is written. */
int mgu_unet_backward(mgu_ctx* ctx, const void* dlogits_dev, void* flat_grad_dev, void* hip_stream);
/* ---- backward building blocks: ONE stage of loss.backward() each, through exactly the launchers mgu_unet_backward uses,
 *      on caller-provided NHWC fp32 tensors (kernel-level parity tests against float64; also usable on their own) ------
 * Weight gradient of Conv2d(k=1|3, pad=k/2) (unet_encoder.py:7-8): dw[co][ci][r][s] = sum_{b,y,x} dz[b,y,x,co] *
 * in[b, y+r-k/2, x+s-k/2, ci].  in: (B,H,W,ld_in), ld_in % 4 == 0, channels [Cin, ld_in) zero; dz: (B,H,W,ceil4(Cout)) with
 * zero pad columns; dw_oihw_dev: (Cout,Cin,k,k), every element written. */
int mgu_conv2d_wgrad_nhwc(mgu_ctx* ctx, const void* in_dev, int ld_in, const void* dz_dev, int B, int H, int W, int Cin, int Cout,
                          int ksize, void* dw_oihw_dev, void* hip_stream);
/* Data gradient of the same layer: din[b,y,x,ci] = sum dz[b, y-r+k/2, x-s+k/2, co] * w[co][ci][r][s];
 * dz as above, din_dev: (B,H,W,ld_out >= Cin). */
int mgu_conv2d_dgrad_nhwc(mgu_ctx* ctx, const void* dz_dev, const void* w_oihw_dev, int B, int H, int W, int Cin, int Cout, int ksize,
                          void* din_dev, int ld_out, void* hip_stream);
/* ConvTranspose2d(Cin, Cout, 2, stride 2) (unet_decoder.py:25,36): in (B,H,W,Cin); dout = channels [c_off, c_off+Cout) of a
 * (B,2H,2W,ld_d) tensor; dw_iohw_dev (Cin,Cout,2,2); dbias_dev (Cout) or NULL. */
int mgu_conv_transpose2x2_wgrad_nhwc(mgu_ctx* ctx, const void* in_dev, const void* dout_dev, int ld_d, int c_off, int B, int H, int W,
                                     int Cin, int Cout, void* dw_iohw_dev, void* dbias_dev, void* hip_stream);
int mgu_conv_transpose2x2_dgrad_nhwc(mgu_ctx* ctx, const void* dout_dev, int ld_d, int c_off, const void* w_iohw_dev, int B, int H,
                                     int W, int Cin, int Cout, void* din_dev /* (B,H,W,Cin) */, void* hip_stream);
/* Train-mode BatchNorm2d (eps 1e-5, momentum 0.1, unet_encoder.py:12-13) + ReLU over an (M, C) view: batch mean / 1/sqrt(biased
 * var + eps) to mean_dev / invstd_dev, running stats updated in place (unbiased var), y = relu(bn(z)) with pitch ld_y. */
int mgu_bn_relu_train_nhwc(mgu_ctx* ctx, const void* z_dev, const void* gamma_dev, const void* beta_dev, int64_t M, int C, void* y_dev,
                           int ld_y, void* mean_dev, void* invstd_dev, void* run_mean_dev, void* run_var_dev, void* hip_stream);
/* ... and its backward: dy (M, ld_dy) -> dz (M, C) dense, dgamma, dbeta (C), dbias = column sums of dz (the conv bias in front
 * of the BatchNorm: analytically 0).  The ReLU mask is recomputed from z. */
int mgu_bn_relu_backward_nhwc(mgu_ctx* ctx, const void* dy_dev, int ld_dy, const void* z_dev, const void* gamma_dev, const void* beta_dev,
                              const void* mean_dev, const void* invstd_dev, int64_t M, int C, void* dz_dev, void* dgamma_dev,
                              void* dbeta_dev, void* dbias_dev, void* hip_stream);
/* MaxPool2d(2,2) backward ACCUMULATED into dskip (the skip tensor also receives the decoder-side gradient): y (B,H,W,ld_y) the
 * pooled tensor's input, dpool (B,H/2,W/2,C) dense, dskip (B,H,W,ld_d) += routed gradient (first maximum wins, as aten). */
int mgu_maxpool2x2_backward_nhwc(mgu_ctx* ctx, const void* y_dev, int ld_y, const void* dpool_dev, void* dskip_dev, int ld_d, int B, int H,
                                 int W, int C, void* hip_stream);

/* torch.optim.Adam.step() with L2 weight decay folded into the gradient (train_segmentation.py:96):
 * g = grad_scale*grad + wd*p; m,v moments; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  step t >= 1.
 * grad_scale lets the caller fold the 1/world_size of an all-reduce SUM into the update. */
int mgu_adam_step(mgu_ctx* ctx, void* flat_param_dev, const void* flat_grad_dev, void* exp_avg_dev, void* exp_avg_sq_dev,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, void* hip_stream);

/* torch.optim.SGD(lr, momentum, weight_decay).step() on the flat buffers -- the other optimizer branch of the reference's train loop
 * (scripts/train_segmentation.py:97-98, selected by configs/training.yaml:5-6; dampening 0, no Nesterov): g = grad_scale*grad + wd*p;
 * momentum != 0: buf = (step == 1 ? g : momentum*buf + g), p -= lr*buf; momentum == 0: p -= lr*g (momentum_buf_dev may be NULL). */
int mgu_sgd_step(mgu_ctx* ctx, void* flat_param_dev, const void* flat_grad_dev, void* momentum_buf_dev, int64_t n, float lr,
                 float momentum, float weight_decay, int step, float grad_scale, void* hip_stream);

/* ---- gradient exchange of the data-parallel train step: RCCL over xGMI (the reference is single-process and has no
 *      collective; BASELINE configs[4] adds exactly this one, SURVEY sections 5 / 8b / 8e) -------------------------------
 * librccl is bound at run time inside libmgunet.so.  The host only moves 128 bytes: rank 0 calls mgu_comm_get_unique_id,
 * hands the bytes to every rank through its launcher's store, and each rank calls mgu_comm_init_rank on its own ctx
 * (= ncclCommInitRank on the ctx's device).  The communicator belongs to the ctx (freed by mgu_comm_destroy /
 * mgu_destroy). */
#define MGU_COMM_ID_BYTES 128
int mgu_comm_get_unique_id(void* id_out /* host, MGU_COMM_ID_BYTES */);
int mgu_comm_init_rank(mgu_ctx* ctx, const void* id /* host, MGU_COMM_ID_BYTES */, int rank, int world_size);
int mgu_comm_destroy(mgu_ctx* ctx);
void* mgu_comm_handle(mgu_ctx* ctx);     /* the ncclComm_t of this ctx, or NULL */
int mgu_comm_world_size(mgu_ctx* ctx);   /* 1 without a communicator */
/* In-place MEAN over the ranks of flat_grad_dev (n fp32, device) on hip_stream: one ncclAllReduce(ncclAvg).
 * rccl_comm: an ncclComm_t of the caller, or NULL for the ctx's own communicator. */
int mgu_allreduce_grads(mgu_ctx* ctx, void* flat_grad_dev, int64_t n, void* rccl_comm, void* hip_stream);
/* mgu_unet_backward + the gradient exchange, overlapped: the flat gradient is mean-all-reduced in buckets (>= 4 MB, in
 * the order backward finishes them: decoder first, encoder last) on the ctx's own communicator stream while the remaining
 * layers are still being differentiated; when the call returns, work enqueued on hip_stream afterwards (mgu_adam_step)
 * sees the averaged gradient.  Needs mgu_comm_init_rank. */
int mgu_unet_backward_allreduce(mgu_ctx* ctx, const void* dlogits_dev, void* flat_grad_dev, void* hip_stream);

/* One-shot request: the NEXT mgu_unet_forward on this ctx also writes the per-patch means of the shallowest decoder
 * feature -- exactly what mgu_patch_mean(decoder_feats[0], ...) returns, (B*nph*npw, init_features) fp32 -- to out_dev.
 * In eval mode with <= 4 classes the means and the final 1x1 conv share ONE pass over the feature map (it is read once
 * instead of twice: both consumers are bandwidth bound).  The node features of the 'full forward' (SURVEY 8a row L3).
 * out_dev == NULL cancels a pending request (e.g. after a forward that failed validation). */
int mgu_unet_request_patch_mean(mgu_ctx* ctx, int patch, void* out_dev);

/* ---- patch graph: replaces preprocessing/graph_construction/patch_graph_construction.py:49-102 -- */
/* ---- GAT training (the reference puts the graph branch's parameters in the optimizer, scripts/train_end_to_end.py:219-226, and
 * differentiates through GraphAttentionLayer.forward with loss.backward(), :478; eval-mode dropout here, train-mode below) ----------
 * DEVICE: transpose of a CSR-by-target: rowptr_src (N+1) / eid_src (E) list, per SOURCE node and in target-CSR order, the positions
 * of its out-edges in col[]; tgt_of_edge (E) is the target row of every edge.  Stable radix sort: every sum of the backward has a
 * fixed order.  Static per graph: build once, reuse for every step. */
int mgu_csr_transpose_device(mgu_ctx* ctx, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E, int num_nodes,
                             int32_t* rowptr_src_dev, int32_t* eid_src_dev, int32_t* tgt_of_edge_dev, void* hip_stream);
/* Backward of mgu_gat_layer_forward (same arguments; the forward's intermediates are recomputed from X, W, a): given dout
 * (N, heads*Fout_head if concat else Fout_head) writes dW (heads*Fout_head, Fin), da (heads, 2*Fout_head) and, if dX_dev is not
 * NULL, dX (N, Fin).  Includes the gradient through the graph-wide max of graph_attention.py:86 (torch.max()'s backward: to the
 * arg-max edge, evenly over ties).  heads*Fout_head <= 256. */
int mgu_gat_layer_backward(mgu_ctx* ctx, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                           const int32_t* rowptr_src_dev, const int32_t* eid_src_dev, const int32_t* tgt_of_edge_dev,
                           const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                           int concat, float alpha, const void* dout_dev, void* dX_dev, void* dW_dev, void* da_dev, void* hip_stream);
/* ---- GAT in TRAIN mode with dropout (model/gat/graph_attention.py:97: nn.Dropout on every head's attention coefficients; :160:
 * nn.Dropout on the layer output; default p = 0.1, configs/model.yaml:18).  The random draw is an explicit MASK (0 or 1 / (1 - p)):
 *   edge_mask (E, heads) fp32 in the CSR-by-target edge order of col[] (head h's mask of edge k at [k * heads + h]); NULL = none
 *   out_mask  (N, heads*Fout_head if concat else Fout_head) fp32; NULL = none
 * so that the same draw can be fed to the reference (a patched nn.Dropout: tests/golden/gat_dropout.npz).  mgu_dropout_mask fills a
 * mask from the library's own counter-based generator (Philox-4x32-10: element i of stream `stream` under `seed` depends on
 * (seed, stream, i) only).  Forward: one GEMM [Wh | s | t], the per-graph max, then one wavefront per target row; the backward is
 * mgu_gat_layer_backward with the masks applied where the forward applies them.  heads*Fout_head <= 256. */
int mgu_dropout_mask(mgu_ctx* ctx, unsigned long long seed, unsigned long long stream, int64_t n, float p, void* mask_dev, void* hip_stream);
int mgu_gat_layer_forward_train(mgu_ctx* ctx, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                                const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                                int concat, float alpha, const void* edge_mask_dev, const void* out_mask_dev, void* out_dev, void* hip_stream);
int mgu_gat_layer_backward_train(mgu_ctx* ctx, const void* X_dev, int N, int Fin, const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                                 const int32_t* rowptr_src_dev, const int32_t* eid_src_dev, const int32_t* tgt_of_edge_dev,
                                 const int32_t* graph_ptr_dev, int num_graphs, const void* W_dev, const void* a_dev, int heads, int Fout_head,
                                 int concat, float alpha, const void* edge_mask_dev, const void* out_mask_dev, const void* dout_dev,
                                 void* dX_dev, void* dW_dev, void* da_dev, void* hip_stream);

/* HOST routine (index maps are tiny and static per image size).  Emits the COO edge_index in the
 * reference's exact order (:77-92) into coo[0..E) (sources) and coo[E..2E) (targets), and the
 * CSR-by-target (rowptr[N+1], col[E]) that keeps each target's sources in COO order.  Any output
 * pointer may be NULL.  *E_out = 2*(nph*(npw-1)+(nph-1)*npw); nodes = nph*npw with ceil-div grid. */
int mgu_patch_graph_build(int H, int W, int patch, int64_t* coo, int32_t* rowptr, int32_t* col,
                          int64_t* E_out, int* nph_out, int* npw_out);
/* HOST: stable COO(2,E int64) -> CSR-by-target for an arbitrary graph (order[k] = COO position). */
int mgu_coo_to_csr(const int64_t* coo, int64_t E, int num_nodes, int32_t* rowptr, int32_t* col);
/* DEVICE: the same for a graph that already lives in device memory (the reference's forward takes any (2, E) int64 edge_index,
 * model/gat/graph_attention.py:40-58): stable sort by target, so each target's sources keep their COO order.  *status_dev (device
 * int) becomes non-zero if an id lies outside [0, num_nodes) -- torch's indexing raises IndexError there; the caller decides when
 * to look at it (one synchronisation per NEW graph, none per forward).  E < 2^31. */
int mgu_coo_to_csr_device(mgu_ctx* ctx, const int64_t* coo_dev, int64_t E, int num_nodes, int32_t* rowptr_dev, int32_t* col_dev,
                          int* status_dev, void* hip_stream);
/* Node features of the 'full forward' (SURVEY 8a row L3): mean over each patch x patch window of an
 * NHWC feature map, zero padded bottom/right as image_to_patches does (:26-47).
 * out_dev: (B*nph*npw, C) fp32. */
int mgu_patch_mean(mgu_ctx* ctx, const void* feat_dev, int feat_dtype /* MGU_DTYPE_* */, int B, int H, int W, int C,
                   int patch, void* out_dev, void* hip_stream);

/* ---- GAT: replaces model/gat/graph_attention.py:40-118, 150-160 (one MultiHeadGATLayer, eval) -- */
/* X_dev (N,Fin) fp32, Fin % 4 == 0; CSR by target on device (rowptr int32[N+1], col int32[E]);
 * W_dev (heads*Fout_head, Fin) = the heads' W.weight stacked; a_dev (heads, 2*Fout_head) = a.weight
 * stacked.  graph_ptr_dev int32[num_graphs+1] gives the node range of each graph of a block-diagonal
 * batch (the reference's exp(e - max(e)) at :86 is per graph and per head); NULL = one graph.
 * concat=1: out (N, heads*Fout_head) = cat of ELU(head) (:155); concat=0: out (N,Fout_head) = mean (:158). */
int mgu_gat_layer_forward(mgu_ctx* ctx, const void* X_dev, int N, int Fin,
                          const int32_t* rowptr_dev, const int32_t* col_dev, int64_t E,
                          const int32_t* graph_ptr_dev, int num_graphs,
                          const void* W_dev, const void* a_dev, int heads, int Fout_head,
                          int concat, float alpha, void* out_dev, void* hip_stream);

/* Steady-state form: the weight-only preparation (W^T a rows, MFMA-fragment-order W^T or the GEMM panel) runs once per weight
 * version, and a layer call is 2 launches when Fin <= Fout_head (st + per-graph max, then gather + aggregate + linear + ELU) or 3
 * otherwise (GEMM [Wh | s | t], per-graph max, row gather).  has_edges: whether the graphs the handle will be used on have edges
 * (a graph without edges -- e.g. a one-node region graph -- takes the gather schedule, whose rows come out exactly 0). */
typedef struct mgu_gat_weights mgu_gat_weights;
int mgu_gat_prepare(mgu_ctx* ctx, const void* W_dev, const void* a_dev, int heads, int Fout_head, int Fin, int has_edges,
                    mgu_gat_weights** out, void* hip_stream);
void mgu_gat_release(mgu_ctx* ctx, mgu_gat_weights* w);
int mgu_gat_layer_forward_prepared(mgu_ctx* ctx, const mgu_gat_weights* w, const void* X_dev, int N, const int32_t* rowptr_dev,
                                   const int32_t* col_dev, int64_t E, const int32_t* graph_ptr_dev, int num_graphs, int concat,
                                   float alpha, void* out_dev, void* hip_stream);

/* ---- MinCut stage of the patch-graph branch: replaces
 *      model/graph_partition/mincut_refinement.py:30-52 (edge weights), :55-160 (normalized-cut loss), :188-205
 *      (softmax of the segment logits + loss), scripts/train_end_to_end.py:356 (hard labels) ---------------------- */
/* w[e] = exp(-|f_src - f_tgt|^2 / 2) for the reference's COO int64 (2,E) edge list (row 0 sources, row 1 targets),
 * in edge order (MinCutRefinement.compute_edge_weights_for_ncut). */
int mgu_ncut_edge_weights(mgu_ctx* ctx, const float* feats_dev, int N, int D, const int64_t* edge_index_dev, int64_t E,
                          float* w_dev, void* hip_stream);
/* loss = sum_k cut_k / assoc_k (segments with assoc_k <= 1e-8 skipped), feats (N,D) fp32, CSR BY SOURCE
 * (rowptr int32[N+1], col int32[E] = targets: the reference sums the degree over the source index, :96).
 * assign_dev (N,K): segment logits (assign_is_logits = 1: softmax over K written to soft_dev, first-arg-max labels to
 * hard_dev if not NULL -- MinCutRefinement.forward) or soft assignments (0: normalized_cut_loss called directly;
 * soft_dev / hard_dev unused).  1 <= K <= 16.  loss_dev: one float. */
int mgu_ncut_forward(mgu_ctx* ctx, const float* feats_dev, int N, int D, const int32_t* rowptr_src_dev,
                     const int32_t* col_tgt_dev, int64_t E, const float* assign_dev, int K, int assign_is_logits,
                     float* soft_dev, int32_t* hard_dev, float* loss_dev, void* hip_stream);
/* loss.backward() through that loss (scripts/train_end_to_end.py:348-356 with :472-479; the reference differentiates through
 * the edge weights as well, mincut_refinement.py:79).  soft_dev (N,K): the assignments the forward used (its soft_dev, or its
 * assign_dev when that already held probabilities); CSR BY SOURCE as in the forward plus CSR BY TARGET (rowptr_tgt, col = sources:
 * mgu_coo_to_csr_device of the unflipped list, or mgu_csr_transpose_device).  gloss_dev: upstream gradient of the loss (one
 * float, NULL = 1); gsoft_dev (N,K) or NULL: upstream gradient of the returned soft assignments (logits only).  Writes
 * dassign_dev (N,K) -- w.r.t. the logits when assign_is_logits, else w.r.t. the probabilities -- and dfeats_dev (N,D) (may be
 * NULL).  Every element is written; both are gathers (no atomics: bitwise reproducible).  D <= 1024, 1 <= K <= 16. */
int mgu_ncut_backward(mgu_ctx* ctx, const float* feats_dev, int N, int D, const int32_t* rowptr_src_dev, const int32_t* col_tgt_dev,
                      const int32_t* rowptr_tgt_dev, const int32_t* col_src_dev, int64_t E, const float* soft_dev, int K,
                      int assign_is_logits, const float* gloss_dev, const float* gsoft_dev, float* dassign_dev, float* dfeats_dev,
                      void* hip_stream);
/* dz[i] = y[i] > 0 ? dy[i] : 0 -- the ReLU of the MLP segment predictor (scripts/train_end_to_end.py:59-63) in its backward. */
int mgu_relu_backward(mgu_ctx* ctx, const float* dy_dev, const float* y_dev, int64_t n, float* dz_dev, void* hip_stream);

/* ---- Region stage + fusion of the e2e forward: replaces scripts/train_end_to_end.py:366-373
 *      (label-mean pooling), :403-421 (region embedding -> patches -> nearest upsample) and the concat of
 *      FeatureFusion.forward (model/fusion_detection/feature_fusion.py:78,145-150) ------------------------------------ */
/* out (B*K, D)[b*K + k] = mean of feats (B*Np, D) rows of image b whose hard label is k; zeros for an empty segment.
 * D % 4 == 0, D <= 1024. */
int mgu_region_mean_pool(mgu_ctx* ctx, const float* feats_dev, const int32_t* hard_dev, int B, int Np, int D, int K,
                         float* out_dev, void* hip_stream);
/* out NHWC (B,H,W,Cu+D): channels [0,Cu) = fu_nhwc (B,H,W,Cu) (Cu may be 0), channels [Cu,Cu+D) = the region embedding
 * (B*K, D) of the segment of the patch the pixel maps to under torch's 'nearest' interpolation of the (nph, npw) grid
 * to (H, W): label = hard[b][min(floor(y*nph/H), nph-1)][min(floor(x*npw/W), npw-1)].  Cu, D multiples of 4. */
int mgu_region_fuse_nhwc(mgu_ctx* ctx, const float* fu_nhwc_dev, int Cu, const float* region_emb_dev, const int32_t* hard_dev,
                         int B, int H, int W, int nph, int npw, int K, int D, float* out_nhwc_dev, void* hip_stream);

/* ---- auxiliary losses of the training loops: forward values, one device float each ---------------------------
 * Streaming reductions in double precision with a fixed summation order (bitwise reproducible); every tensor is addressed by
 * ELEMENT strides so NCHW and NHWC storage both work without a copy. */
/* TVLoss.forward (scripts/train_end_to_end.py:73-89): weight * (sum (x[y+1]-x[y])^2 / ((H-1) W) + sum (x[x+1]-x[x])^2 / (H (W-1))) / B
 * over x (B,C,H,W), element (n,c,y,x) at x_dev[n*xs_n + c*xs_c + y*xs_h + x*xs_w]. */
int mgu_tv_loss(mgu_ctx* ctx, const void* x_dev, int B, int C, int H, int W, int64_t xs_n, int64_t xs_c, int64_t xs_h, int64_t xs_w,
                float weight, float* loss_dev, void* hip_stream);
/* dice_loss (scripts/train_segmentation.py:29-40): softmax over the classes, one-hot target, Dice with additive smoothing per
 * (image, class), 1 - mean.  logits element (b, c, pixel p) at logits_dev[b*ls_n + c*ls_c + p*ls_p]; labels int64 (B, HW);
 * num_classes <= 8.  A label outside [0, num_classes) (F.one_hot raises) is reported like mgu_cross_entropy's. */
int mgu_dice_loss(mgu_ctx* ctx, const void* logits_dev, const int64_t* labels_dev, int B, int64_t HW, int num_classes, int64_t ls_n,
                  int64_t ls_c, int64_t ls_p, float smooth, float* loss_dev, void* hip_stream);
/* FeatureConsistencyLoss.forward (model/unet/feature_loss.py:88-125), the (B, N, D) / (B, N) form: f_unet, f_graph (B,N,D) fp32
 * contiguous, y (B,N) fp32 (the reference casts y.float()): mean_b sum_n [ y d^2 + (1-y) relu(margin - sqrt(d^2 + 1e-8))^2 ]. */
int mgu_feature_consistency_loss(mgu_ctx* ctx, const void* f_unet_dev, const void* f_graph_dev, const void* y_dev, int B, int N, int D,
                                 float margin, float* loss_dev, void* hip_stream);
/* ---- gradients of the differentiable auxiliary losses ---------------------------------------------------------------------------
 * The reference obtains them from autograd (loss.backward(): scripts/train_segmentation.py:133, scripts/train_end_to_end.py:478).
 * Each entry multiplies the gradient of the scalar loss by grad_scale and, when grad_scale_dev is not NULL, by that device float
 * too (an autograd grad_output never has to visit the host).  EllipticalShapeLoss has no gradient w.r.t. its input: it is a
 * function of the arg-max pixel COORDINATES (shape_loss.py:61-98) -- and the reference loop pins loss_shape to 0 (:287). */
/* d TVLoss / dx into dx_dev, element (n,c,y,x) at dx_dev[n*ds_n + c*ds_c + y*ds_h + x*ds_w]. */
int mgu_tv_loss_backward(mgu_ctx* ctx, const void* x_dev, int B, int C, int H, int W, int64_t xs_n, int64_t xs_c, int64_t xs_h,
                         int64_t xs_w, float weight, float grad_scale, const float* grad_scale_dev, void* dx_dev, int64_t ds_n,
                         int64_t ds_c, int64_t ds_h, int64_t ds_w, void* hip_stream);
/* d dice_loss / d logits (through the softmax), element (b, c, p) at dlogits_dev[b*ds_n + c*ds_c + p*ds_p]; accumulate != 0 ADDS to
 * what is there -- the trainer's `loss_ce + loss_dice` (scripts/train_segmentation.py:126-133) is mgu_cross_entropy followed by this
 * call on the same dlogits.  loss_dev (optional) receives the loss value of the same pass. */
int mgu_dice_loss_backward(mgu_ctx* ctx, const void* logits_dev, const int64_t* labels_dev, int B, int64_t HW, int num_classes,
                           int64_t ls_n, int64_t ls_c, int64_t ls_p, float smooth, float grad_scale, const float* grad_scale_dev,
                           void* dlogits_dev, int64_t ds_n, int64_t ds_c, int64_t ds_p, int accumulate, float* loss_dev, void* hip_stream);
/* d FeatureConsistencyLoss / d f_unet and / d f_graph ((B,N,D) contiguous each; either may be NULL). */
int mgu_feature_consistency_loss_backward(mgu_ctx* ctx, const void* f_unet_dev, const void* f_graph_dev, const void* y_dev, int B, int N,
                                          int D, float margin, float grad_scale, const float* grad_scale_dev, void* d_f_unet_dev,
                                          void* d_f_graph_dev, void* hip_stream);
/* Synchronise the stream and report (MGU_ERR_INVALID) a label outside [0, num_classes) met by mgu_dice_loss[_backward] on this
 * context since the last check: the place F.one_hot would have raised. */
int mgu_loss_sync_check(mgu_ctx* ctx, void* hip_stream);
/* EllipticalShapeLoss.forward (model/unet/shape_loss.py:17-180).  _masks: the object_masks_list form, all masks of the batch
 * stacked (num_objects, H, W) uint8 (non-zero = object pixel).  _probs: the form without masks -- per image, the pixels whose
 * arg-max class is 1 are ONE object (:61-98); probabilities (B,C,H,W) at probs_dev[b*ps_n + c*ps_c + p*ps_p].  Objects under 10
 * pixels are skipped; the loss is the mean over the processed objects of mean_pixels (p^T (cov + eps I)^-1 p - 1)^2, 0 if none. */
int mgu_elliptical_shape_loss_masks(mgu_ctx* ctx, const uint8_t* masks_dev, int num_objects, int H, int W, float epsilon, float* loss_dev,
                                    void* hip_stream);
int mgu_elliptical_shape_loss_probs(mgu_ctx* ctx, const void* probs_dev, int B, int num_classes, int H, int W, int64_t ps_n, int64_t ps_c,
                                    int64_t ps_p, float epsilon, float* loss_dev, void* hip_stream);

/* ---- resize / gather building blocks of FeatureFusion (model/fusion_detection/feature_fusion.py:43-162) ----------------------------
 * F.interpolate(mode='bilinear', align_corners=False) (:69-76, :140-144) of an NHWC fp32 map (B,Hi,Wi,C) with pixel pitch ld_in into
 * channels [c_off, c_off + C) of a (B,Ho,Wo,ld_out) buffer -- i.e. straight into its slice of the fused tensor. */
int mgu_resize_bilinear_nhwc(mgu_ctx* ctx, const void* in_dev, int ld_in, int B, int Hi, int Wi, int C, void* out_dev, int ld_out, int c_off,
                             int Ho, int Wo, void* hip_stream);
/* The per-region branch (:84-138): out[pixel][c_off + d] = table[ids[pixel]][d]; an id outside [0, R) gives zeros (the reference
 * leaves such pixels of its zero-initialised map untouched).  table (R, D) fp32, ids int64 (npix). */
int mgu_region_map_gather_nhwc(mgu_ctx* ctx, const float* table_dev, int R, int D, const int64_t* ids_dev, int64_t npix, float* out_dev,
                               int ld_out, int c_off, void* hip_stream);

/* ---- input / output pipeline around the network: byte and integer work reproduced exactly ---------------------------
 * The reference does these steps on the host with cv2 / PIL / torchvision.  Images are HWC uint8 in device memory.
 * ImagePreprocessor.preprocess (preprocessing/image_preprocessing/image_preprocess.py:26-31, 57-85): [BGR -> RGB | grey -> RGB] ->
 * torchvision Resize on a PIL image = PIL's antialiased BILINEAR resample in 8-bit fixed point (horizontal pass, then vertical, 22-bit
 * coefficients) -> ToTensor (/255) -> Normalize.  mean3 / std3: HOST arrays of 3 floats.  out element (c, y, x) at
 * out_dev[c*os_c + y*os_h + x*os_w] (fp32): CHW as the reference returns it, or an NHWC slot of a batch. */
int mgu_preprocess_image_u8(mgu_ctx* ctx, const uint8_t* img_dev, int Hs, int Ws, int channels /* 1 | 3 */, int bgr, int H, int W,
                            const float* mean3, const float* std3, void* out_dev, int64_t os_c, int64_t os_h, int64_t os_w, void* hip_stream);
/* preprocess_mask (:87-126): cv2.resize(INTER_NEAREST) (source index floor(dst * src/dst), clamped), np.clip to [0, num_classes-1], int64. */
int mgu_preprocess_mask_u8(mgu_ctx* ctx, const uint8_t* mask_dev, int Hs, int Ws, int H, int W, int num_classes, int64_t* out_dev,
                           void* hip_stream);
/* EdgeDetector.sobel_edges (preprocessing/graph_feature_processing/edge_detection.py:14-44), kernel size 3: RGB -> grey (14-bit fixed
 * point), Sobel x / y with reflect-101 borders, magnitude / max * 255 in double, truncated to uint8.  rgb (H,W,3) -> out (H,W). */
int mgu_sobel_edges_u8(mgu_ctx* ctx, const uint8_t* rgb_dev, int H, int W, uint8_t* out_dev, void* hip_stream);
/* HistogramEqualizer.equalize_histogram_rgb (histogram_equalization.py:13-35): RGB -> YUV, equalizeHist on Y, YUV -> RGB. */
int mgu_equalize_hist_rgb_u8(mgu_ctx* ctx, const uint8_t* rgb_dev, int H, int W, uint8_t* out_dev, void* hip_stream);
/* image_to_patches(map).mean(...) (scripts/graph_refinement.py:97-104): mean over each patch x patch window of a uint8 HWC map, zero
 * padded bottom / right; per_channel = 0: one value per patch over all channels, 1: one per channel.  out (nph*npw, 1 | channels). */
int mgu_patch_mean_u8(mgu_ctx* ctx, const uint8_t* img_dev, int H, int W, int channels, int patch, int per_channel, float* out_dev,
                      void* hip_stream);
/* postprocess_segmentation (scripts/infer_segmentation.py:20-51, :123): class labels int64 -> colour map uint8 (npix, 3) through a
 * palette (num_classes, 3) the caller provides (the reference's BGR list), labels outside [0, num_classes) stay black; and, if
 * labels_u8_dev != NULL, the uint8 label map written next to it. */
int mgu_colorize_labels(mgu_ctx* ctx, const int64_t* labels_dev, int64_t npix, const uint8_t* palette_dev, int num_classes, uint8_t* vis_dev,
                        uint8_t* labels_u8_dev, void* hip_stream);

/* ---- per-channel building blocks of the DetectionHead: model/fusion_detection/detection_head.py ---- */
/* y[m][c] = act(scale[c] * x[m][c] + shift[c]) over an (M, C) NHWC view with row pitches ldx / ldy (floats); scale / shift
 * may be NULL (1 / 0); act 0 = none (the BatchNorm2d that follows a ReLU, :33-38), 1 = ReLU, 2 = sigmoid (:101,104).
 * C, ldx, ldy multiples of 4. */
int mgu_channel_affine_nhwc(mgu_ctx* ctx, const float* x_dev, int ldx, int64_t M, int C, const float* scale_dev,
                            const float* shift_dev, int act, float* y_dev, int ldy, void* hip_stream);
/* out[c] = sum over the M rows of x[m][c]: AdaptiveAvgPool2d((1,1)) (:39) per image is this sum / (H W). 4 <= C <= 1024. */
int mgu_channel_sum_nhwc(mgu_ctx* ctx, const float* x_dev, int ldx, int64_t M, int C, float* out_dev, void* hip_stream);

/* out (B, C)[b] = column sums of image b of x (B, M, C) (row pitch ldx): AdaptiveAvgPool2d over a batch in one call. */
int mgu_channel_sum_images_nhwc(mgu_ctx* ctx, const float* x_dev, int ldx, int B, int64_t M, int C, float* out_dev, void* hip_stream);

/* ---- introspection for bench.py / profiles ------------------------------------------------------ */
/* FLOPs (2*MAC, convolutions only) of one U-Net forward over B images: SURVEY 8d table. */
double mgu_unet_flops(mgu_ctx* ctx, int B, int H, int W);
/* FLOPs the matrix cores actually EXECUTE for the same forward: the fp32 3x3 layers with Cin % 16 == 0 run as
 * Winograd F(2x2,3x3) -- 16 multiplies per 2x2 output tile and (cin, cout) pair instead of 36 (csrc/wino_f32.hip);
 * equal to mgu_unet_flops when that path is off (bf16 storage, MGU_NO_WINOGRAD=1). */
double mgu_unet_mfma_flops(mgu_ctx* ctx, int B, int H, int W);
/* Time the conv/GEMM kernels of the LAST mgu_unet_forward with HIP events on the launch stream:
 * enable before the forward, read after.  Adds event records only (no syncs) while enabled. */
int mgu_profile_enable(mgu_ctx* ctx, int on);
/* Per kernel family, summed over the launches recorded since mgu_profile_enable(ctx, 1) (U-Net forward / backward convolutions,
 * GAT kernels): time between HIP events recorded on the launch stream right around each launch, algorithmic FLOPs (2*MAC of the
 * operator) and the FLOPs actually issued on the matrix pipe (`pipe`: 0 fp32 MFMA, 1 bf16 MFMA, -1 none).  `name` is the kernel's
 * name as rocprofv3 --kernel-trace prints it (template arguments included where two instantiations are used).  A read CONSUMES the
 * records: with profiling left on, the next read covers the launches since this one. */
typedef struct {
  const char* name;
  double ms, flops_alg, flops_mfma;
  int launches, pipe;
} mgu_kernel_stat;
int mgu_profile_read_kernels(mgu_ctx* ctx, mgu_kernel_stat* out, int cap, int* n_out);
int mgu_profile_read(mgu_ctx* ctx, double* conv_ms, int* conv_launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MGUNET_H */
