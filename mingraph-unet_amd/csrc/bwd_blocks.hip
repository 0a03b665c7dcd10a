// Backward building blocks behind the C-ABI: each entry point runs ONE stage of loss.backward()
// (scripts/train_segmentation.py:133) through exactly the launchers mgu_unet_backward uses (mgunet_train.hip), on
// caller-provided tensors.  They exist so that every backward kernel can be checked in isolation against a float64
// reference on fixed (x, dz) -- where nothing is ill-conditioned -- instead of only through a whole train step whose
// BatchNorm + ReLU + MaxPool chain amplifies rounding (tests/test_gpu_backward_kernels.py).
// Kernel selection follows the context's switches (MGU_NO_WINO_WGRAD, MGU_NO_WGRAD_HALO, MGU_NO_THIN_WGRAD,
// MGU_NO_WINO_DGRAD, MGU_NO_WINOGRAD read at mgu_create), so a test reaches every variant.
#include <algorithm>

#include "ctx.h"

using namespace mgu;
using namespace mgud;

namespace {

// scale / shift of the train-mode forward from the batch statistics: y = relu(scale * z + shift)
__global__ void fold_batch_stats_kernel(const float* gamma, const float* beta, const float* mean, const float* invstd, float* scale,
                                        float* shift, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    const float sc = gamma[i] * invstd[i];
    scale[i] = sc;
    shift[i] = beta[i] - mean[i] * sc;
  }
}

struct Scratch {
  float *dwp, *dgp, *wug;
  size_t dwp_floats;
  double *red, *sums;
};

int get_scratch(mgu_ctx* c, size_t panel_floats, size_t dgp_floats, size_t wug_floats, int Cmax, Scratch* out) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t dwp_floats = std::max(panel_floats, (size_t)12 << 20);   // room for the atomics-free kernels' partial panels
  const size_t o_dwp = take(dwp_floats * 4), o_dgp = take(std::max<size_t>(dgp_floats, 64) * 4), o_wug = take((wug_floats + 64) * 4);
  const size_t o_sums = take(sizeof(double) * 2 * (size_t)std::max(Cmax, 64) + 64);
  int rc = ensure(c, &c->gws, &c->gws_bytes, off);
  if (rc) return rc;
  const size_t need = chan_reduce_work_bytes(std::max(Cmax, 64));
  if (c->redws_bytes < need) {   // the slots must be zero between reductions: a fresh allocation is cleared once
    if ((rc = ensure(c, &c->redws, &c->redws_bytes, need))) return rc;
    HIPCHK(c, hipMemset(c->redws, 0, need));
  }
  char* g = (char*)c->gws;
  out->dwp = (float*)(g + o_dwp), out->dgp = (float*)(g + o_dgp), out->wug = (float*)(g + o_wug);
  out->dwp_floats = dwp_floats;
  out->sums = (double*)(g + o_sums);
  out->red = (double*)c->redws;
  return MGU_OK;
}

}  // namespace

extern "C" {

int mgu_conv2d_wgrad_nhwc(mgu_ctx* c, const void* in_dev, int ld_in, const void* dz_dev, int B, int H, int W, int Cin, int Cout,
                          int ksize, void* dw_oihw_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!in_dev || !dz_dev || !dw_oihw_dev || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (ksize != 1 && ksize != 3))
    return fail(c, MGU_ERR_INVALID, "bad conv2d_wgrad args (ksize must be 1 or 3)");
  const int Cp = rup(Cin, 4), N = rup(Cout, 4);
  if (ld_in < Cp || (ld_in & 3)) return fail(c, MGU_ERR_INVALID, "ld_in must be a multiple of 4 and >= Cin rounded up to 4");
  if ((int64_t)B * H * W >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int K = ksize * ksize * Cp, Kp = rup(K, 32);
  Scratch sc;
  int rc = get_scratch(c, (size_t)rup(N, 128) * Kp, 0, 0, 64, &sc);
  if (rc) return rc;
  WgradDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.z = (const float*)dz_dev, d.ldz = N, d.zoff = 0;   // dz rows are padded to a multiple of 4 channels (zeros)
  d.in = (const float*)in_dev, d.ldin = ld_in, d.inoff = 0, d.Cp = Cp;
  d.KS = ksize;
  d.M = B * H * W, d.H = H, d.W = W;
  d.N = N, d.K = K, d.Kp = Kp;
  d.dw = sc.dwp, d.dw_capacity = sc.dwp_floats;
  HIPCHK(c, launch_wgrad_f32(d, s));
  HIPCHK(c, launch_unpack_conv_grad(sc.dwp, d.groups, (size_t)d.N * d.Kp, (float*)dw_oihw_dev, Cout, Cin, Cp, ksize, Kp, s));
  return MGU_OK;
}

int mgu_conv2d_dgrad_nhwc(mgu_ctx* c, const void* dz_dev, const void* w_oihw_dev, int B, int H, int W, int Cin, int Cout, int ksize,
                          void* din_dev, int ld_out, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!dz_dev || !w_oihw_dev || !din_dev || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (ksize != 1 && ksize != 3))
    return fail(c, MGU_ERR_INVALID, "bad conv2d_dgrad args (ksize must be 1 or 3)");
  if (ld_out < Cin) return fail(c, MGU_ERR_INVALID, "ld_out %d < Cin %d", ld_out, Cin);
  if ((int64_t)B * H * W >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int Cop = rup(Cout, 4), Kd = ksize * ksize * Cop, Kpd = rup(Kd, 32);
  const bool wino = c->tn.wino_dgrad && c->tn.use_wino && ksize == 3 && Cop % 16 == 0;
  Scratch sc;
  int rc = get_scratch(c, 0, (size_t)rup(Cin, 128) * Kpd, wino ? wino_u_floats(Cin, Cop) : 0, 64, &sc);
  if (rc) return rc;
  IgemmDesc d;
  memset(&d, 0, sizeof d);
  d.tn = &c->tn;
  d.in = (const float*)dz_dev, d.w = sc.dgp, d.out = (float*)din_dev;
  d.M = B * H * W, d.H = H, d.W = W;
  d.Cp = Cop, d.ldin = Cop, d.KS = ksize, d.K = Kd, d.Kp = Kpd;
  d.N = Cin, d.ldout = ld_out;
  if (wino) d.wu = sc.wug;
  if (wino_applicable(d)) {
    HIPCHK(c, launch_pack_wino_w((const float*)w_oihw_dev, sc.wug, Cin, Cout, Cop, 1, c->tn.wino_prec, s));
  } else {
    HIPCHK(c, hipMemsetAsync(sc.dgp, 0, (size_t)rup(Cin, 128) * Kpd * sizeof(float), s));   // panel rows are padded to 128
    HIPCHK(c, launch_pack_dgrad_w((const float*)w_oihw_dev, sc.dgp, Cout, Cin, Cop, ksize, Kpd, s));
  }
  HIPCHK(c, launch_igemm_f32(d, s));
  return MGU_OK;
}

int mgu_conv_transpose2x2_wgrad_nhwc(mgu_ctx* c, const void* in_dev, const void* dout_dev, int ld_d, int c_off, int B, int H, int W,
                                     int Cin, int Cout, void* dw_iohw_dev, void* dbias_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!in_dev || !dout_dev || !dw_iohw_dev || B < 1 || H < 1 || W < 1 || Cin < 4 || (Cin & 3) || Cout < 4 || (Cout & 3) || Cout > 1024)
    return fail(c, MGU_ERR_INVALID, "bad convT_wgrad args (Cin, Cout multiples of 4)");
  if (ld_d < c_off + Cout || (ld_d & 3) || (c_off & 3)) return fail(c, MGU_ERR_INVALID, "ld_d / c_off must be multiples of 4, ld_d >= c_off + Cout");
  if ((int64_t)B * H * W * 4 >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "4*B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int Kt = 4 * Cout, Kpt = rup(Kt, 32);
  Scratch sc;
  int rc = get_scratch(c, (size_t)rup(Cin, 128) * Kpt, 0, 0, Cout, &sc);
  if (rc) return rc;
  // the roles swap (mgu_unet_backward): Z = the layer's INPUT (M, Cin), A = 2x2 stride-2 gather of d(out)
  WgradDesc g;
  memset(&g, 0, sizeof g);
  g.tn = &c->tn;
  g.z = (const float*)in_dev, g.ldz = Cin, g.in = (const float*)dout_dev, g.ldin = ld_d, g.inoff = c_off, g.Cp = Cout, g.KS = 2;
  g.M = B * H * W, g.H = H, g.W = W, g.Hs = 2 * H, g.Ws = 2 * W;
  g.N = Cin, g.K = Kt, g.Kp = Kpt, g.dw = sc.dwp, g.dw_capacity = sc.dwp_floats;
  HIPCHK(c, launch_wgrad_f32(g, s));
  HIPCHK(c, launch_unpack_convt_grad(sc.dwp, g.groups, (size_t)g.N * g.Kp, (float*)dw_iohw_dev, Cin, Cout, Kpt, s));
  if (dbias_dev)
    HIPCHK(c, launch_colsum((const float*)dout_dev + c_off, ld_d, (int64_t)B * H * W * 4, Cout, sc.red, (float*)dbias_dev, s));
  return MGU_OK;
}

int mgu_conv_transpose2x2_dgrad_nhwc(mgu_ctx* c, const void* dout_dev, int ld_d, int c_off, const void* w_iohw_dev, int B, int H,
                                     int W, int Cin, int Cout, void* din_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!dout_dev || !w_iohw_dev || !din_dev || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 4 || (Cout & 3))
    return fail(c, MGU_ERR_INVALID, "bad convT_dgrad args (Cout a multiple of 4)");
  if (ld_d < c_off + Cout || (ld_d & 3) || (c_off & 3)) return fail(c, MGU_ERR_INVALID, "ld_d / c_off must be multiples of 4, ld_d >= c_off + Cout");
  if ((int64_t)B * H * W * 4 >= (1ll << 31)) return fail(c, MGU_ERR_INVALID, "4*B*H*W must be < 2^31");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int Kt = 4 * Cout, Kpt = rup(Kt, 32);
  Scratch sc;
  const size_t dgp_floats = std::max((size_t)rup(Cin, 128) * Kpt, convt_x3_dgrad_floats(Cin, Cout));
  int rc = get_scratch(c, 0, dgp_floats, 0, 64, &sc);
  if (rc) return rc;
  IgemmDesc q;
  memset(&q, 0, sizeof q);
  q.tn = &c->tn;
  q.in = (const float*)dout_dev + c_off, q.w = sc.dgp, q.out = (float*)din_dev, q.M = B * H * W, q.H = H, q.W = W, q.Cp = Cout,
  q.ldin = ld_d;
  q.KS = 2, q.K = Kt, q.Kp = Kpt, q.N = Cin, q.ldout = Cin, q.Hout = 2 * H, q.Wout = 2 * W;
  q.wu = sc.dgp;
  if (c->tn.convt_dgrad_x3 && convt_x3_dgrad_applicable(q)) {   // the forward layer's three-piece kernel in its gather mode
    HIPCHK(c, launch_pack_convt_x3_dgrad((const float*)w_iohw_dev, sc.dgp, Cin, Cout, s));
  } else {
    q.wu = nullptr;
    HIPCHK(c, hipMemsetAsync(sc.dgp, 0, (size_t)rup(Cin, 128) * Kpt * sizeof(float), s));
    HIPCHK(c, launch_pack_convt_dgrad_w((const float*)w_iohw_dev, sc.dgp, Cin, Cout, Kpt, s));
  }
  HIPCHK(c, launch_igemm_f32(q, s));
  return MGU_OK;
}

int mgu_bn_relu_train_nhwc(mgu_ctx* c, const void* z_dev, const void* gamma_dev, const void* beta_dev, int64_t M, int C, void* y_dev,
                           int ld_y, void* mean_dev, void* invstd_dev, void* run_mean_dev, void* run_var_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!z_dev || !gamma_dev || !beta_dev || !y_dev || !mean_dev || !invstd_dev || !run_mean_dev || !run_var_dev || M < 2 || C < 4 ||
      (C & 3) || C > 1024 || ld_y < C || (ld_y & 3))
    return fail(c, MGU_ERR_INVALID, "bad bn_relu_train args (4 <= C <= 1024, C and ld_y multiples of 4, M >= 2)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  Scratch sc;
  int rc = get_scratch(c, 0, 2 * (size_t)C, 0, C, &sc);   // dgp holds the folded scale / shift
  if (rc) return rc;
  float *tscale = sc.dgp, *tshift = sc.dgp + C;
  HIPCHK(c, launch_bn_stats((const float*)z_dev, C, M, C, sc.red, sc.sums, s));
  HIPCHK(c, launch_bn_finalize(sc.sums, sc.sums + C, M, 1e-5f, 0.1f, (const float*)gamma_dev, (const float*)beta_dev, (float*)mean_dev,
                               (float*)invstd_dev, tscale, tshift, (float*)run_mean_dev, (float*)run_var_dev, C, s));
  HIPCHK(c, launch_bn_apply_relu((const float*)z_dev, tscale, tshift, (float*)y_dev, ld_y, M, C, s));
  return MGU_OK;
}

int mgu_bn_relu_backward_nhwc(mgu_ctx* c, const void* dy_dev, int ld_dy, const void* z_dev, const void* gamma_dev, const void* beta_dev,
                              const void* mean_dev, const void* invstd_dev, int64_t M, int C, void* dz_dev, void* dgamma_dev,
                              void* dbeta_dev, void* dbias_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!dy_dev || !z_dev || !gamma_dev || !beta_dev || !mean_dev || !invstd_dev || !dz_dev || !dgamma_dev || !dbeta_dev || !dbias_dev ||
      M < 2 || C < 4 || (C & 3) || C > 1024 || ld_dy < C || (ld_dy & 3))
    return fail(c, MGU_ERR_INVALID, "bad bn_relu_backward args (4 <= C <= 1024, C and ld_dy multiples of 4, M >= 2)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  Scratch sc;
  int rc = get_scratch(c, 0, 2 * (size_t)C, 0, C, &sc);
  if (rc) return rc;
  // forward scale/shift (the ReLU mask is recomputed from z): scale = gamma*invstd, shift = beta - mean*scale
  float *tscale = sc.dgp, *tshift = sc.dgp + C;
  hipLaunchKernelGGL(fold_batch_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)gamma_dev, (const float*)beta_dev,
                     (const float*)mean_dev, (const float*)invstd_dev, tscale, tshift, C);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, launch_bn_bwd_reduce((const float*)dy_dev, ld_dy, tscale, tshift, (const float*)z_dev, C, (const float*)mean_dev,
                                 (const float*)invstd_dev, M, C, sc.red, sc.sums, (float*)dbeta_dev, (float*)dgamma_dev, s));
  HIPCHK(c, launch_bn_bwd_apply((const float*)dy_dev, ld_dy, tscale, tshift, (const float*)z_dev, (const float*)mean_dev,
                                (const float*)invstd_dev, (const float*)gamma_dev, sc.sums, M, C, (float*)dz_dev, sc.red,
                                (float*)dbias_dev, s));
  return MGU_OK;
}

int mgu_maxpool2x2_backward_nhwc(mgu_ctx* c, const void* y_dev, int ld_y, const void* dpool_dev, void* dskip_dev, int ld_d, int B, int H,
                                 int W, int Cc, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!y_dev || !dpool_dev || !dskip_dev || B < 1 || H < 2 || W < 2 || Cc < 4 || (Cc & 3) || ld_y < Cc || (ld_y & 3) || ld_d < Cc || (ld_d & 3))
    return fail(c, MGU_ERR_INVALID, "bad maxpool_backward args (C, ld_y, ld_d multiples of 4, H,W >= 2)");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_maxpool2_bwd_add((const float*)y_dev, ld_y, (const float*)dpool_dev, (float*)dskip_dev, ld_d, B, H, W, Cc,
                                    (hipStream_t)hip_stream));
  return MGU_OK;
}

}  // extern "C"
