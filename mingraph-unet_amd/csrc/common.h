// Internal declarations shared by the HIP translation units of libmgunet.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgu {

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor.  D[m][n] = act(scale[n] * sum_k A(m,k) * Wp[n][k] + shift[n])
//   A(m,k) is gathered on the fly from an NHWC activation tensor:
//     k = tap*Cp + c, tap = (r*KS + s); A = in[img, oy + r - KS/2, ox + s - KS/2, c]  (0 outside)
//   Wp is the packed weight panel [Np][Kp] (k contiguous), zero padded.
// out_mode 0: D row m -> out[m*ldout + coff + n]                       (conv3x3 / 1x1 / linear)
// out_mode 1: ConvTranspose2d k2 s2 pixel-shuffle: n = (dy*2+dx)*ct_cout + co,
//             m = (img, y, x) over the INPUT grid H x W,
//             -> out[((img*Hout + 2y+dy)*Wout + 2x+dx)*ldout + coff + co]
// ---------------------------------------------------------------------------------------------
// Kernel-selection switches of ONE context (mgu_ctx::tn, filled from the MGU_* environment in mgu_create).  They ride in
// the launch descriptors, so two contexts of a process never see each other's settings.
struct Tuning {
  bool first_mfma = true;   // MGU_NO_FIRST_MFMA=1: the first convolution on the VALU kernel (conv3x3_first_kernel) instead of the matrix cores (A/B)
  bool use_halo = true;     // MGU_NO_HALO=1: generic gather kernel instead of the LDS-halo conv kernel (A/B)
  bool halo_tps3 = true;    // MGU_HALO_TPS1=1: one tap per barrier on the N <= 32 halo tile too (A/B)
  int halo_max_ppb = 16;    // MGU_HALO_PPB=n: patches a halo workgroup walks (1 = no persistence)
  bool use_wino = true;     // MGU_NO_WINOGRAD=1: direct kernels for the fp32 3x3 layers
  int wino_mode = -1;       // MGU_WINO_MODE=1: force the 32-channel work split on every layer (A/B)
  int wino_prec = 1;        // MGU_WINO_PREC: 1 = three exact bf16 pieces per fp32 operand on the bf16 MFMA (default),
                            //                0 = fp32 MFMA operands
  bool wino_cp_narrow = true;   // MGU_WINO_CP_NARROW=0: N <= 32 layers stay on wino3x3_f32_kernel<1,1> (A/B)
  bool wino_yfast = false;  // MGU_WINO_YFAST=1: Winograd / halo workgroups walk their patches y fastest inside an image (A/B)
  bool convt_frag = true;   // MGU_NO_CONVT_FRAG=1: ConvTranspose on the generic tile kernel instead of convt_x3.hip's kernels (A/B)
  bool wino_deep = true;    // MGU_NO_WINO_DEEP=1: one chunk of load lead on the narrow Winograd layers too (A/B)
  bool wino_cp = true;      // MGU_NO_WINO_CP=1: the four-components-per-wave kernel instead of the component-pair split (A/B)
  int wino_rounds = 1;      // MGU_WINO_ROUNDS / MGU_WINO_PPB_CAP: persistence of the Winograd workgroups
  int wino_ppb_cap = 32;
  bool wgrad_halo = true;   // MGU_NO_WGRAD_HALO=1
  bool wino_wgrad = true;   // MGU_NO_WINO_WGRAD=1
  bool convt_dgrad_x3 = true;   // MGU_NO_CONVT_DGRAD_X3=1: ConvTranspose data gradient on the generic fp32 tile kernel
  bool wgrad_x3 = true;     // MGU_NO_WGRAD_X3=1: Winograd weight gradient on the fp32 MFMA instead of the three-piece bf16 products
  bool wgrad_thin = true;   // MGU_NO_THIN_WGRAD=1
  bool wino_dgrad = true;   // MGU_NO_WINO_DGRAD=1
  bool gat_fused = true;    // MGU_NO_GAT_FUSED=1
  bool wino_ures = true;    // MGU_NO_WINO_URES=1: the 32-input-channel narrow layers reload their weight pieces every chunk (A/B)
  bool wino_prio = false;   // MGU_WINO_PRIO=1: s_setprio 1 for waves 4-7 of the component-pair Winograd kernels (A/B)
  bool wino_asm_narrow = true;   // MGU_WINO_ASM_NARROW=0: the assembly form only for the wide layers (A/B)
  int wino_asm = 1;         // MGU_WINO_ASM=0: the C++ component-pair kernels instead of their hand-scheduled assembly forms (wino_asm.hip; bitwise
                            // equal results, A/B and fallback); n > 1: timing-only variant n - 1 of a GEN_WINO_VARIANTS=1 build (never shipped)
};
const Tuning& default_tuning();
// The >64 KB dynamic-LDS opt-in is a per-DEVICE function attribute: set it once per (kernel, device).
hipError_t ensure_dyn_lds(const void* func, size_t bytes, bool (&done)[64]);

struct IgemmDesc {
  const Tuning* tn;    // nullptr = default_tuning()
  const float* in;
  const float* w;
  const float* wu;     // optional: Winograd-transformed 3x3 weights (launch_pack_wino_w); enables wino_f32.hip
  const float* scale;  // may be nullptr (== 1)
  const float* shift;  // may be nullptr (== 0)
  float* out;
  int M;        // rows = B*H*W
  int H, W;     // spatial grid that M enumerates
  int Cp;       // channels gathered per tap (multiple of 4)
  int ldin;     // channel pitch of `in` (elements between pixels)
  int KS;       // 1 or 3
  int K;        // KS*KS*Cp
  int Kp;       // row pitch of w (multiple of 32)
  int N;        // valid output columns
  int ldout;    // channel pitch of out
  int coff;     // channel offset inside the out pixel
  int relu;
  int out_mode;
  int ct_cout;     // out_mode 1: Cout (N == 4*Cout)
  int Hout, Wout;  // out_mode 1: output grid (>= 2H, 2W)
  // optional fused MaxPool2d(2) of the output (Winograd kernel only): pool[(img, y/2, x/2)*ldpool + n], floor semantics
  float* pool;
  int ldpool;
  // optional (Winograd kernel, training forward): per-channel sum / sum of squares of the stored output accumulated into the
  // row-per-workgroup double accumulator [STAT_ROWS][2 * N] (train_kernels.hip), folded by launch_bn_finalize_slots
  double* stat_slots;
  // optional split epilogue (out_mode 0): columns n >= split_n go to out2[m*ld2 + (n - split_n)] (0 = off)
  int split_n;
  float* out2;
  int ld2;
};

inline const Tuning& tun(const IgemmDesc& d) { return d.tn ? *d.tn : default_tuning(); }
hipError_t launch_igemm_f32(const IgemmDesc& d, hipStream_t s);
int wino_grid_blocks(const IgemmDesc& d);   // wino_f32.hip: workgroups of the Winograd launch for d (= accumulator rows of its statistics)
bool halo_pool_fusable(const IgemmDesc& d, int dtype);   // the halo conv kernel will run: MaxPool2d(2) can ride in its epilogue
hipError_t launch_igemm_bf16(const IgemmDesc& d, hipStream_t s);
const char* igemm_kernel_name(const IgemmDesc& d, int dtype);
// convt_x3.hip: fp32 ConvTranspose2d(k2,s2) on the bf16 matrix cores with exact three-way operand splits (IgemmDesc::wu =
// fragment-ordered weight pieces)
size_t convt_x3_floats(int Cin, int Cout);
hipError_t launch_pack_convt_x3(const float* w, float* Wx, int Cin, int Cout, hipStream_t s);
bool convt_x3_applicable(const IgemmDesc& d);
hipError_t launch_convt_x3(const IgemmDesc& d, hipStream_t s);
// convt_bf16.hip: the bf16-storage ConvTranspose2d(k2,s2) on fragment-ordered bf16 weights (IgemmDesc::wu), 16-byte transposed stores
size_t convt_bf16f_floats(int Cin, int Cout);
hipError_t launch_pack_convt_bf16f(const float* w, float* Wf, int Cin, int Cout, hipStream_t s);
bool convt_bf16f_applicable(const IgemmDesc& d);
hipError_t launch_convt_bf16f(const IgemmDesc& d, hipStream_t s);
// the same kernel as the layer's data gradient (KS = 2 gather descriptors whose d.wu holds launch_pack_convt_x3_dgrad's panel)
size_t convt_x3_dgrad_floats(int Cin, int Cout);
hipError_t launch_pack_convt_x3_dgrad(const float* w, float* Wx, int Cin, int Cout, hipStream_t s);
bool convt_x3_dgrad_applicable(const IgemmDesc& d);
hipError_t launch_convt_x3_dgrad(const IgemmDesc& d, hipStream_t s);
  // the kernel family launch_igemm_* will pick (profiling records)  // in / w / out point to bf16, sizes in elements
// elementwise.hip: first convolution (<= 4 input channels on the packed NHWC4 input), VALU + scalar-cache weights
hipError_t launch_pack_first_w(const float* w, float* wf, int Cout, int Cin, hipStream_t s);
// first_mfma.hip: the same layer on the bf16 matrix cores (Cin <= 3, Cout == 32)
size_t first_mfma_floats();
bool first_mfma_applicable(int dtype, int Cin, int Cp, int Cout, int ldout, int coff, int64_t H, int64_t W);
hipError_t launch_pack_first_mfma(const float* w, float* wfm, int Cout, int Cin, hipStream_t s);
hipError_t launch_first_mfma(int dtype, const void* in, const float* wfm, const float* scale, const float* shift, void* out, int B, int H,
                             int W, int ldout, int coff, int relu, hipStream_t s);
hipError_t launch_first_mfma_direct(int dtype, const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int cin, const float* wfm,
                                    const float* scale, const float* shift, void* out, int B, int H, int W, int ldout, int coff, int relu,
                                    hipStream_t s);
bool first_conv_applicable(int dtype, int Cin, int Cp, int Cout, int ldout, int coff);
hipError_t launch_first_conv(int dtype, const void* in, const float* wf, const float* scale, const float* shift, void* out, int B, int H, int W,
                             int Cin, int Cout, int ldout, int coff, int relu, hipStream_t s);
// wino_f32.hip: Winograd F(2x2,3x3) for fp32 3x3 layers with Cp % 16 == 0
size_t wino_u_floats(int Cout, int Cp);
hipError_t launch_pack_wino_w(const float* w, float* U, int Cout, int Cin, int Cp, int dgrad, int prec, hipStream_t s);
// several Winograd weight sets in one launch (wino_f32.hip): w (Cout, Cin, 3, 3) [dgrad: roles swapped, see pack_wino_w_kernel]
// kind 0: a Winograd set.  The small per-layer forms ride in the same launch (pack_small.h): kind 1 = first-conv weights (w, U = wf,
// Cout, Cin); 2 = three-piece ConvTranspose fragments (w, U = Wx, Cin, Cout, dgrad = forward / data gradient); 3 = bias tile (w = bias,
// U = shift, Cout = C, Cin = reps); 4 = direct data-gradient panel (w, U = wp, Cout, Cin, Cp = Cop, Np = Kp, dgrad = KS)
enum { PACK_WINO = 0, PACK_FIRST_W = 1, PACK_CONVT_X3 = 2, PACK_BIAS_TILE = 3, PACK_DGRAD_W = 4, PACK_FIRST_MFMA = 5 };   // 5: (w, U = wfm, Cout, Cin)
struct WinoPackItem {
  const float* w;
  float* U;
  int Cout, Cin, Cp, Np, dgrad;
  unsigned blk0;   // Np (kind 0), blk0: filled by the launcher
  int kind;
  int reserved;    // no padding bytes: repack_weights compares the tables with memcmp to skip the upload
};
static_assert(sizeof(WinoPackItem) == 48, "WinoPackItem must have no padding");
constexpr int WINO_PACK_MAX = 64;
struct WinoPackBatch {
  int n, prec;
  unsigned total_blocks;
  WinoPackItem it[WINO_PACK_MAX];
};
bool wino_pack_batch_prepare(WinoPackBatch& b);   // fills Np / blk0 / total_blocks; false if an item cannot be packed
hipError_t launch_pack_wino_w_multi(const WinoPackBatch* batch_dev, unsigned total_blocks, hipStream_t s);
bool wino_applicable(const IgemmDesc& d);
hipError_t launch_wino_f32(const IgemmDesc& d, hipStream_t s);
// wino_asm.hip: the assembly form of wino3x3_cp_kernel<2> (bitwise equal results)
bool wino_asm_applicable(const IgemmDesc& d);
hipError_t launch_wino_cp_asm(const IgemmDesc& d, hipStream_t s);

// wgrad_f32.hip:  Dw[n][k] += sum_m Z[m][n] * A(m,k)   (A = the forward kernels' im2col gather)
struct WgradDesc {
  const Tuning* tn;  // nullptr = default_tuning()
  const float* z;   // Z[m][n] at z[m*ldz + zoff + n]
  int ldz, zoff;
  const float* in;  // gather source (NHWC), channels [inoff, inoff+Cp) of a pixel with pitch ldin
  int ldin, inoff, Cp;
  int KS;           // 1 | 3 (pad 1) | 2 (stride-2 2x2 gather from an Hs x Ws grid: ConvTranspose2d)
  int M, H, W;      // rows enumerate (img, y, x) over H x W
  int Hs, Ws;       // KS == 2 source grid
  int N, K, Kp;     // Dw panel [>=N][Kp]
  float* dw;        // [groups][>=N][Kp] partial panels; the launcher sets `groups`.  Plain stores, every element (n < N, k < K)
                    // of every partial panel written (no zeroing needed, no atomics); the caller sums the panels in order
  size_t dw_capacity;  // floats available at dw
  int groups;          // set by the launcher
  int rows_per_split;  // set by the launcher
};
inline const Tuning& tun(const WgradDesc& d) { return d.tn ? *d.tn : default_tuning(); }
hipError_t launch_wgrad_f32(WgradDesc& d, hipStream_t s);
// wino_wgrad_f32.hip: Winograd F(3x3,2x2) weight gradient (Cp % 64 == 0, N % 64 == 0); same partial-panel output as the halo kernel
bool wino_wgrad_applicable(const WgradDesc& d);
hipError_t launch_wino_wgrad_f32(WgradDesc& d, hipStream_t s);
// wgrad_thin.hip: the first 3x3 conv (Cin 3) and the 1x1 head: HBM-bound streaming kernels, same partial-panel output
bool wgrad_thin_applicable(const WgradDesc& d);
hipError_t launch_wgrad_thin(WgradDesc& d, hipStream_t s);

// train_kernels.hip
// Per-channel reductions meet in a table of double rows [rows][2 * C]; every workgroup adds into a row of its OWN (row =
// blockIdx.x: one adder per element, onto zero), and the reader folds the rows in a fixed order and clears them -- so a sum is
// bitwise reproducible from run to run (64 shared slots with several adders each were not: the order of double atomics moved
// the last bit of a BatchNorm statistic about once in a thousand steps).  STAT_ROWS bounds the grid of the Winograd kernels
// that accumulate statistics in their epilogue (one round of <= 2 x 256 workgroups + rounding).
constexpr int STAT_ROWS = 576;
constexpr int CHAN_REDUCE_ROWS = 512;   // workgroups (= table rows) of a per-channel reduction: 2 per CU, each streaming with 4 loads in flight per thread
size_t chan_reduce_work_bytes(int Cmax);
hipError_t launch_bn_stats(const float* z, int ldz, int64_t M, int C, double* work, double* sums, hipStream_t s);
hipError_t launch_bn_finalize(const double* sum, const double* sumsq, int64_t M, float eps, float momentum,
                              const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                              float* shift, float* run_mean, float* run_var, int C, hipStream_t s);
hipError_t launch_bn_finalize_slots(double* slots, int nrows, double* sums, int64_t M, float eps, float momentum, const float* gamma,
                                    const float* beta, float* mean, float* invstd, float* scale, float* shift, float* run_mean,
                                    float* run_var, int C, hipStream_t s);
hipError_t launch_bn_apply_relu(const float* z, const float* scale, const float* shift, float* y, int ldy, int64_t M, int C,
                                hipStream_t s);
hipError_t launch_bn_apply_relu_pool(const float* z, const float* scale, const float* shift, float* y, int ldy, float* pooled, int B, int H,
                                     int W, int C, hipStream_t s);
hipError_t launch_bn_bwd_reduce(const float* dy, int lddy, const float* fwd_scale, const float* fwd_shift, const float* z, int ldz,
                                const float* mean, const float* invstd, int64_t M, int C, double* work, double* sums,
                                float* dbeta, float* dgamma, hipStream_t s);
hipError_t launch_bn_bwd_apply(const float* dy, int lddy, const float* fwd_scale, const float* fwd_shift, const float* z, const float* mean,
                               const float* invstd, const float* gamma, const double* sums, int64_t M, int C, float* dz,
                               double* work, float* dbias, hipStream_t s);
hipError_t launch_bn_bwd_apply_deferred(const float* dy, int lddy, const float* fsc, const float* fsh, const float* z, const float* mean,
                                        const float* invstd, const float* gamma, const double* sums, int64_t M, int C, float* dz,
                                        double* work, int* rows, hipStream_t s);
hipError_t launch_colsum(const float* z, int ldz, int64_t M, int C, double* work, float* out, hipStream_t s);
hipError_t launch_maxpool2_bwd_add(const float* y, int ldy, const float* dpool, float* dskip, int ldd, int B, int H, int W,
                                   int C, hipStream_t s);
hipError_t launch_zero_pad_region(float* buf, int ld, int coff, int C, int B, int H, int W, int h2, int w2, hipStream_t s);
hipError_t launch_ce(const float* logits, const int64_t* labels, int64_t M, int C, long long ignore_index, float grad_scale,
                     float* dlogits, int ldd, double* acc, int* err_word, float* loss_out, hipStream_t s);
hipError_t launch_pack_dgrad_w(const float* w, float* wp, int Cout, int Cin, int Cop, int KS, int Kp, hipStream_t s);
hipError_t launch_pack_convt_dgrad_w(const float* w, float* wp, int Cin, int Cout, int Kp, hipStream_t s);
hipError_t launch_unpack_conv_grad(const float* dwp, int groups, size_t panel_stride, float* g, int Cout, int Cin, int Cp, int KS,
                                   int Kp, hipStream_t s, double* fold_slots = nullptr, int fold_rows = 0, int fold_n = 0, float* fold_out = nullptr);
hipError_t launch_unpack_convt_grad(const float* dwp, int groups, size_t panel_stride, float* g, int Cin, int Cout, int Kp, hipStream_t s);
hipError_t launch_sgd(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd, int step, float grad_scale,
                      hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                       float wd, int step, float grad_scale, hipStream_t s);

// elementwise.hip
// dtype: 0 = fp32, 1 = bf16 storage (MGU_DTYPE_*); void* buffers hold that type
hipError_t launch_pack_input(const float* x, void* out, int dtype, int B, int C, int Cp, int H, int W, int64_t sn, int64_t sc,
                             int64_t sh, int64_t sw, hipStream_t s);
hipError_t launch_maxpool2(const void* in, int ldin, void* out, int dtype, int B, int H, int W, int C, hipStream_t s);
bool patch_mean_head_fusable(int dtype, int C, int ncls);
hipError_t launch_patch_mean(const void* feat, int dtype, float* out, int B, int H, int W, int C, int patch, hipStream_t s,
                             const float* head_w = nullptr, const float* head_b = nullptr, float* logits = nullptr, int ncls = 0);
hipError_t launch_pack_conv_w(const float* w_oihw, void* wp, int dtype, int Cout, int Cin, int Cp, int KS, int Kp, hipStream_t s);
hipError_t launch_pack_convt_w(const float* w_iohw, void* wp, int dtype, int Cin, int Cout, int Kp, hipStream_t s);
hipError_t launch_bn_fold(const float* bias, const float* gamma, const float* beta, const float* mean, const float* var,
                          float eps, float* scale, float* shift, int C, hipStream_t s);
hipError_t launch_bias_tile(const float* bias, float* shift, int C, int reps, hipStream_t s);
hipError_t launch_conv1x1_head(const void* in, int dtype, int ldin, int C, const float* w, const float* bias, float* out,
                               int ldout, int ncls, int64_t npix, hipStream_t s);
hipError_t launch_argmax(const float* logits, int64_t npix, int C, int64_t* pred, hipStream_t s);

// gat.hip
hipError_t launch_gat_wa_rows(const float* W, const float* a, float* panel, int row0, int heads, int Fh, int Fin, int Kp, hipStream_t s);
// gat_fused.hip: aggregate-first path (Fin <= F')
bool gat_fused_applicable(int Fin, int heads, int Fh, int64_t E);
size_t gat_fused_scratch_floats(int Fin, int heads, int Fh);
hipError_t launch_gat_prep(const float* W, const float* a, float* wa, unsigned* Wx, int heads, int Fh, int Fin, hipStream_t s);
hipError_t launch_gat_stmax(const float* x, const float* wa, int N, int Fin, int heads, const int32_t* rowptr, const int32_t* col,
                            const int32_t* gp, int G, float alpha, float* st, int32_t* node_graph, unsigned long long* gmax, int gstride,
                            unsigned gen, hipStream_t s);
hipError_t launch_gat_fused(const float* x, int Fin, const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* gp, int G,
                            const unsigned long long* gmax, const unsigned* Wx, int N, int heads, int Fh, int concat, float alpha, float* out,
                            int gstride, unsigned gen, hipStream_t s);
hipError_t launch_gat_node_graph(const int32_t* gp, int G, int nodes_per_graph, int N, int32_t* node_graph, hipStream_t s);
hipError_t launch_gat_edge_max(const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* node_graph, int N,
                               int heads, float alpha, unsigned long long* gmax_enc, int gstride, unsigned gen, hipStream_t s);
hipError_t launch_gat_aggregate(const float* wh, int P, const float* st, const int32_t* rowptr, const int32_t* col,
                                const int32_t* node_graph, const unsigned long long* gmax_enc, int N, int64_t E, int heads, int Fh,
                                int concat, float alpha, float* out, int gstride, unsigned gen, hipStream_t s);

}  // namespace mgu
