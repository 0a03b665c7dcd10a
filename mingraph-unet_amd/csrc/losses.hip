// Auxiliary losses of the training loops, forward values and gradients on the device:
//   mgu_tv_loss                    TVLoss.forward                      scripts/train_end_to_end.py:73-89
//   mgu_dice_loss                  dice_loss                           scripts/train_segmentation.py:29-40
//   mgu_feature_consistency_loss   FeatureConsistencyLoss.forward      model/unet/feature_loss.py:88-125
//   mgu_elliptical_shape_loss_*    EllipticalShapeLoss.forward         model/unet/shape_loss.py:17-180
// All four are streaming reductions (HBM bound).  Every workgroup folds its part in double precision and writes ONE partial
// record (no atomics); a single-workgroup finishing kernel adds the records in a fixed order, so a loss value is bitwise
// reproducible from run to run, and forms the scalar the reference returns.  The reference's per-object Python loop of the shape
// loss (nonzero -> mean -> torch.cov -> torch.inverse -> diag(X S^-1 X^T), an N x N matrix per object!) becomes two passes over
// the masks: raw coordinate moments per object, then the Mahalanobis residuals with the 2 x 2 inverse in closed form.
#include <algorithm>

#include "ctx.h"

namespace mgu {

constexpr int LOSS_BLOCKS = 1024;   // upper bound of reduction workgroups (partial records)

// fold K doubles of every thread of a 256-thread workgroup; thread 0 gets the totals
template <int K>
__device__ __forceinline__ void block_fold(double (&v)[K], double* sh /* [4][K] */) {
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < K; ++k) sh[wave * K + k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = sh[k] + sh[K + k] + sh[2 * K + k] + sh[3 * K + k];
}

// ---- total variation ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tv_partial_kernel(const float* __restrict__ x, int B, int Cc, int H, int W, int64_t sn, int64_t sc,
                                                         int64_t sh_, int64_t sw, double* __restrict__ part) {
  __shared__ double sh[8];
  double v[2] = {0.0, 0.0};
  const int64_t total = (int64_t)B * Cc * H * W;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xw = (int)(i % W);
    int64_t r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % Cc), n = (int)(r / Cc);
    const float* p = x + n * sn + c * sc + y * sh_ + xw * sw;
    const float a = *p;
    if (y + 1 < H) {
      const float d = p[sh_] - a;
      v[0] += (double)d * d;
    }
    if (xw + 1 < W) {
      const float d = p[sw] - a;
      v[1] += (double)d * d;
    }
  }
  block_fold<2>(v, sh);
  if (threadIdx.x == 0) part[blockIdx.x * 2] = v[0], part[blockIdx.x * 2 + 1] = v[1];
}
__global__ void tv_final_kernel(const double* __restrict__ part, int nb, double count_h, double count_w, double weight, double B,
                                float* __restrict__ out) {
  double h = 0, w = 0;
  for (int i = 0; i < nb; ++i) h += part[2 * i], w += part[2 * i + 1];
  // torch divides the fp32 sums; the quotient order is kept: weight * (h / count_h + w / count_w) / B  (:88)
  *out = (float)(weight * (h / count_h + w / count_w) / B);
}

// ---- dice -----------------------------------------------------------------------------------------------------------------
// per (image b, class c): I = sum p_c [y == c], P = sum p_c, T = sum [y == c]
template <int NC>
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           int64_t HW, int Cc, int64_t ls_n, int64_t ls_c, int64_t ls_p,
                                                           double* __restrict__ part, int* __restrict__ err_word) {
  __shared__ double sh[4 * 3 * NC];
  double v[3 * NC];
#pragma unroll
  for (int k = 0; k < 3 * NC; ++k) v[k] = 0.0;
  const int b = blockIdx.y;
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
    const float* p = logits + b * ls_n + i * ls_p;
    float l[NC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      l[c] = c < Cc ? p[c * ls_c] : -INFINITY;
      mx = fmaxf(mx, l[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      l[c] = c < Cc ? expf(l[c] - mx) : 0.f;
      se += l[c];
    }
    const float inv = 1.f / se;
    const long long y = labels[b * HW + i];
    if (y < 0 || y >= Cc) bad = true;     // F.one_hot raises on such a label (:34)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float pc = l[c] * inv;
      v[c] += y == c ? (double)pc : 0.0;
      v[NC + c] += (double)pc;
      v[2 * NC + c] += y == c ? 1.0 : 0.0;
    }
  }
  if (bad && err_word) atomicOr(err_word, 1);
  block_fold<3 * NC>(v, sh);
  if (threadIdx.x == 0)
    for (int k = 0; k < 3 * NC; ++k) part[((size_t)b * gridDim.x + blockIdx.x) * 3 * NC + k] = v[k];
}
template <int NC>
__global__ void dice_final_kernel(const double* __restrict__ part, int B, int nb, int Cc, double smooth, float* __restrict__ out) {
  double acc = 0;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < Cc; ++c) {
      double I = 0, P = 0, T = 0;
      for (int k = 0; k < nb; ++k) {
        const double* p = part + ((size_t)b * nb + k) * 3 * NC;
        I += p[c], P += p[NC + c], T += p[2 * NC + c];
      }
      acc += (2.0 * I + smooth) / (P + T + smooth);   // :39
    }
  *out = (float)(1.0 - acc / ((double)B * Cc));       // :40
}

// ---- feature consistency --------------------------------------------------------------------------------------------------
// one 16-lane group per (b, n) row, D / 4 float4 steps; y as float
__global__ __launch_bounds__(256) void featcons_partial_kernel(const float* __restrict__ fu, const float* __restrict__ fg,
                                                               const float* __restrict__ y, int64_t rows, int D, float margin,
                                                               double* __restrict__ part) {
  __shared__ double sh[4];
  double v[1] = {0.0};
  const int q = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)256 + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * 256) >> 4;
  for (int64_t r = grp; r < rows; r += ngrp) {
    float d2 = 0.f;
    for (int c = 4 * q; c < D; c += 64) {
      const float4 a = *reinterpret_cast<const float4*>(fu + r * D + c), b = *reinterpret_cast<const float4*>(fg + r * D + c);
      const float e0 = a.x - b.x, e1 = a.y - b.y, e2 = a.z - b.z, e3 = a.w - b.w;
      d2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) d2 += __shfl_xor(d2, off);
    if (q == 0) {
      const float yp = y[r];
      const float dist = sqrtf(d2 + 1e-8f);                     // :115
      const float hinge = fmaxf(margin - dist, 0.f);            // :117
      v[0] += (double)(yp * d2 + (1.f - yp) * hinge * hinge);   // :109, :118, :120
    }
  }
  block_fold<1>(v, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}
__global__ void sum_final_kernel(const double* __restrict__ part, int nb, double scale, float* __restrict__ out) {
  double s = 0;
  for (int i = 0; i < nb; ++i) s += part[i];
  *out = (float)(s * scale);
}


// ---- gradients of the differentiable losses (the reference obtains them from autograd: loss.backward() at
// scripts/train_segmentation.py:133, scripts/train_end_to_end.py:478) ---------------------------------------------------------
// Every backward kernel multiplies by gs = grad_scale * (*grad_scale_dev if given): the upstream gradient of the scalar loss,
// as a host number, a device scalar (an autograd grad_output: no host synchronisation) or both.
__device__ __forceinline__ float up_scale(float gs, const float* __restrict__ gs_dev) { return gs_dev ? gs * *gs_dev : gs; }

// d TVLoss / dx (scripts/train_end_to_end.py:84-88): each squared difference feeds its two end points
__global__ __launch_bounds__(256) void tv_bwd_kernel(const float* __restrict__ x, int B, int Cc, int H, int W, int64_t sn, int64_t sc,
                                                     int64_t sh_, int64_t sw, float kh, float kw, float gs, const float* __restrict__ gs_dev,
                                                     float* __restrict__ dx, int64_t dn, int64_t dc, int64_t dh, int64_t dw) {
  const float g = up_scale(gs, gs_dev);
  const int64_t total = (int64_t)B * Cc * H * W;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xw = (int)(i % W);
    int64_t r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % Cc), n = (int)(r / Cc);
    const float* p = x + n * sn + c * sc + y * sh_ + xw * sw;
    const float a = *p;
    float v = 0.f, h = 0.f;
    if (y > 0) v += a - p[-sh_];
    if (y + 1 < H) v -= p[sh_] - a;
    if (xw > 0) h += a - p[-sw];
    if (xw + 1 < W) h -= p[sw] - a;
    dx[n * dn + c * dc + y * dh + xw * dw] = g * (kh * v + kw * h);
  }
}

// dice: q[b][c] = dL/dp_c of a pixel of image b = coef[b][0][c] * [y == c] + coef[b][1][c]
//   L = 1 - mean_{b,c} (2 I + s) / (P + T + s)  ->  dL/dI = -2 / (B C U),  dL/dP = (2 I + s) / (B C U^2),  U = P + T + s
template <int NC>
__global__ void dice_coef_kernel(const double* __restrict__ part, int B, int nb, int Cc, double smooth, float* __restrict__ coef) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * NC) return;
  const int b = t / NC, c = t - b * NC;
  float c0 = 0.f, c1 = 0.f;
  if (c < Cc) {
    double I = 0, P = 0, T = 0;
    for (int k = 0; k < nb; ++k) {
      const double* p = part + ((size_t)b * nb + k) * 3 * NC;
      I += p[c], P += p[NC + c], T += p[2 * NC + c];
    }
    const double U = P + T + smooth, inv = 1.0 / ((double)B * Cc);
    c0 = (float)(-2.0 * inv / U);
    c1 = (float)((2.0 * I + smooth) * inv / (U * U));
  }
  coef[(b * 2 + 0) * NC + c] = c0;
  coef[(b * 2 + 1) * NC + c] = c1;
}
// softmax backward: dlogit_k = p_k (q_k - sum_c q_c p_c); written (or added: the trainer's CE + dice) to dlogits (B*HW, ldd)
template <int NC>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int64_t HW,
                                                       int Cc, int64_t ls_n, int64_t ls_c, int64_t ls_p, const float* __restrict__ coef,
                                                       float gs, const float* __restrict__ gs_dev, float* __restrict__ dlogits,
                                                       int64_t ds_n, int64_t ds_c, int64_t ds_p, int accumulate) {
  const int b = blockIdx.y;
  const float g = up_scale(gs, gs_dev);
  float c0[NC], c1[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) c0[c] = coef[(b * 2 + 0) * NC + c], c1[c] = coef[(b * 2 + 1) * NC + c];
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
    const float* p = logits + b * ls_n + i * ls_p;
    float l[NC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      l[c] = c < Cc ? p[c * ls_c] : -INFINITY;
      mx = fmaxf(mx, l[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      l[c] = c < Cc ? expf(l[c] - mx) : 0.f;
      se += l[c];
    }
    const float inv = 1.f / se;
    const long long y = labels[b * HW + i];
    float q[NC], dot = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      l[c] *= inv;
      q[c] = (y == c ? c0[c] : 0.f) + c1[c];
      dot += q[c] * l[c];
    }
    float* d = dlogits + b * ds_n + i * ds_p;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c < Cc) {
        const float v = g * l[c] * (q[c] - dot);
        d[c * ds_c] = accumulate ? d[c * ds_c] + v : v;
      }
  }
}

// feature consistency: d/df_unet = k (a - b) / B,  k = 2 y - 2 (1 - y) relu(m - dist) / dist;  d/df_graph = -that   (:106-123)
__global__ __launch_bounds__(256) void featcons_bwd_kernel(const float* __restrict__ fu, const float* __restrict__ fg,
                                                           const float* __restrict__ y, int64_t rows, int D, float margin, float gs,
                                                           const float* __restrict__ gs_dev, float* __restrict__ dfu,
                                                           float* __restrict__ dfg) {
  const float g = up_scale(gs, gs_dev);
  const int q = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)256 + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * 256) >> 4;
  for (int64_t r = grp; r < rows; r += ngrp) {
    float d2 = 0.f;
    for (int c = 4 * q; c < D; c += 64) {
      const float4 a = *reinterpret_cast<const float4*>(fu + r * D + c), b = *reinterpret_cast<const float4*>(fg + r * D + c);
      const float e0 = a.x - b.x, e1 = a.y - b.y, e2 = a.z - b.z, e3 = a.w - b.w;
      d2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) d2 += __shfl_xor(d2, off);
    const float yp = y[r];
    const float dist = sqrtf(d2 + 1e-8f);
    const float hinge = fmaxf(margin - dist, 0.f);
    const float k = g * (2.f * yp - 2.f * (1.f - yp) * hinge / dist);
    for (int c = 4 * q; c < D; c += 64) {
      const float4 a = *reinterpret_cast<const float4*>(fu + r * D + c), b = *reinterpret_cast<const float4*>(fg + r * D + c);
      const float4 o = make_float4(k * (a.x - b.x), k * (a.y - b.y), k * (a.z - b.z), k * (a.w - b.w));
      if (dfu) *reinterpret_cast<float4*>(dfu + r * D + c) = o;
      if (dfg) *reinterpret_cast<float4*>(dfg + r * D + c) = make_float4(-o.x, -o.y, -o.z, -o.w);
    }
  }
}

// ---- elliptical shape -----------------------------------------------------------------------------------------------------
// MODE 0: masks (M, H, W) uint8, object = mask m.  MODE 1: probabilities (B, C, H, W) with element strides, object b = the pixels
// whose arg-max class (first maximum, as torch.argmax) is 1.
template <int MODE>
__device__ __forceinline__ bool in_object(const void* __restrict__ src, int obj, int64_t i, int64_t HW, int Cc, int64_t sn, int64_t sc,
                                          int64_t sp) {
  if (MODE == 0) return reinterpret_cast<const uint8_t*>(src)[obj * HW + i] != 0;
  const float* p = reinterpret_cast<const float*>(src) + obj * sn + i * sp;
  float best = p[0];
  int arg = 0;
  for (int c = 1; c < Cc; ++c) {
    const float v = p[c * sc];
    if (v > best) best = v, arg = c;
  }
  return arg == 1;
}
// pass 1: n, sum y, sum x, sum yy, sum xy, sum xx per object (exact in double for any image this library accepts)
template <int MODE>
__global__ __launch_bounds__(256) void shape_moments_kernel(const void* __restrict__ src, int H, int W, int Cc, int64_t sn, int64_t sc,
                                                            int64_t sp, double* __restrict__ part) {
  __shared__ double sh[24];
  double v[6] = {0, 0, 0, 0, 0, 0};
  const int obj = blockIdx.y;
  const int64_t HW = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256)
    if (in_object<MODE>(src, obj, i, HW, Cc, sn, sc, sp)) {
      const double y = (double)(i / W), x = (double)(i % W);
      v[0] += 1.0, v[1] += y, v[2] += x, v[3] += y * y, v[4] += y * x, v[5] += x * x;
    }
  block_fold<6>(v, sh);
  if (threadIdx.x == 0)
    for (int k = 0; k < 6; ++k) part[((size_t)obj * gridDim.x + blockIdx.x) * 6 + k] = v[k];
}
// centroid + inverse of (sample covariance + eps I) per object: obj_par[obj] = {valid, cy, cx, i00, i01, i11}
__global__ void shape_params_kernel(const double* __restrict__ part, int nobj, int nb, double eps, double* __restrict__ obj_par) {
  const int obj = blockIdx.x * blockDim.x + threadIdx.x;
  if (obj >= nobj) return;
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < nb; ++k)
    for (int j = 0; j < 6; ++j) m[j] += part[((size_t)obj * nb + k) * 6 + j];
  double* o = obj_par + (size_t)obj * 8;
  const double n = m[0];
  if (n < 10.0) {   // objects under 10 pixels are skipped (:96, :100, :157)
    o[0] = 0.0;
    return;
  }
  const double cy = m[1] / n, cx = m[2] / n;
  // torch.cov: divisor N - 1 (:131); + eps I (:137)
  const double syy = (m[3] - n * cy * cy) / (n - 1.0) + eps, sxy = (m[4] - n * cy * cx) / (n - 1.0), sxx = (m[5] - n * cx * cx) / (n - 1.0) + eps;
  const double det = syy * sxx - sxy * sxy;
  o[0] = 1.0, o[1] = cy, o[2] = cx, o[3] = sxx / det, o[4] = -sxy / det, o[5] = syy / det, o[6] = n;
}
// pass 2: sum over the object's pixels of (p^T S^-1 p - 1)^2
template <int MODE>
__global__ __launch_bounds__(256) void shape_residual_kernel(const void* __restrict__ src, int H, int W, int Cc, int64_t sn, int64_t sc,
                                                             int64_t sp, const double* __restrict__ obj_par, double* __restrict__ part) {
  __shared__ double sh[4];
  double v[1] = {0.0};
  const int obj = blockIdx.y;
  const double* o = obj_par + (size_t)obj * 8;
  const int64_t HW = (int64_t)H * W;
  if (o[0] != 0.0) {
    const double cy = o[1], cx = o[2], i00 = o[3], i01 = o[4], i11 = o[5];
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256)
      if (in_object<MODE>(src, obj, i, HW, Cc, sn, sc, sp)) {
        const double y = (double)(i / W) - cy, x = (double)(i % W) - cx;
        const double mah = y * (i00 * y + i01 * x) + x * (i01 * y + i11 * x);   // :143
        v[0] += (mah - 1.0) * (mah - 1.0);                                      // :145
      }
  }
  block_fold<1>(v, sh);
  if (threadIdx.x == 0) part[(size_t)obj * gridDim.x + blockIdx.x] = v[0];
}
__global__ void shape_final_kernel(const double* __restrict__ part, const double* __restrict__ obj_par, int nobj, int nb,
                                   float* __restrict__ out) {
  double total = 0;
  int processed = 0;
  for (int obj = 0; obj < nobj; ++obj) {
    const double* o = obj_par + (size_t)obj * 8;
    if (o[0] == 0.0) continue;
    double s = 0;
    for (int k = 0; k < nb; ++k) s += part[(size_t)obj * nb + k];
    total += s / o[6];   // mean over the object's pixels
    ++processed;
  }
  *out = processed ? (float)(total / processed) : 0.f;   // :147, :177
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

namespace {

int loss_scratch(mgu_ctx* c, size_t doubles, double** out) {
  int rc = ensure(c, &c->lossws, &c->lossws_bytes, doubles * sizeof(double));
  if (rc) return rc;
  *out = (double*)c->lossws;
  return MGU_OK;
}
inline int nblocks(int64_t work) { return (int)std::max<int64_t>(1, std::min<int64_t>(LOSS_BLOCKS, (work + 1023) / 1024)); }

template <int MODE>
int shape_loss(mgu_ctx* c, const void* src, int nobj, int H, int W, int Cc, int64_t sn, int64_t sc, int64_t sp, float eps, float* loss,
               hipStream_t s) {
  const int nb = std::min(256, nblocks((int64_t)H * W));
  double* ws;
  int rc = loss_scratch(c, (size_t)nobj * nb * 6 + (size_t)nobj * 8 + (size_t)nobj * nb, &ws);
  if (rc) return rc;
  double *mom = ws, *par = ws + (size_t)nobj * nb * 6, *res = par + (size_t)nobj * 8;
  hipLaunchKernelGGL(shape_moments_kernel<MODE>, dim3(nb, nobj), dim3(256), 0, s, src, H, W, Cc, sn, sc, sp, mom);
  hipLaunchKernelGGL(shape_params_kernel, dim3((nobj + 63) / 64), dim3(64), 0, s, mom, nobj, nb, (double)eps, par);
  hipLaunchKernelGGL(shape_residual_kernel<MODE>, dim3(nb, nobj), dim3(256), 0, s, src, H, W, Cc, sn, sc, sp, par, res);
  hipLaunchKernelGGL(shape_final_kernel, dim3(1), dim3(1), 0, s, res, par, nobj, nb, loss);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

}  // namespace

extern "C" {

int mgu_tv_loss(mgu_ctx* c, const void* x_dev, int B, int Cc, int H, int W, int64_t xs_n, int64_t xs_c, int64_t xs_h, int64_t xs_w,
                float weight, float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!x_dev || !loss_dev || B < 1 || Cc < 1 || H < 2 || W < 2) return fail(c, MGU_ERR_INVALID, "bad tv_loss args (H, W >= 2)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int nb = nblocks((int64_t)B * Cc * H * W);
  double* ws;
  int rc = loss_scratch(c, (size_t)nb * 2, &ws);
  if (rc) return rc;
  hipLaunchKernelGGL(tv_partial_kernel, dim3(nb), dim3(256), 0, s, (const float*)x_dev, B, Cc, H, W, xs_n, xs_c, xs_h, xs_w, ws);
  hipLaunchKernelGGL(tv_final_kernel, dim3(1), dim3(1), 0, s, ws, nb, (double)(H - 1) * W, (double)H * (W - 1), (double)weight, (double)B,
                     loss_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_dice_loss(mgu_ctx* c, const void* logits_dev, const int64_t* labels_dev, int B, int64_t HW, int num_classes, int64_t ls_n,
                  int64_t ls_c, int64_t ls_p, float smooth, float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!logits_dev || !labels_dev || !loss_dev || B < 1 || HW < 1 || num_classes < 1 || num_classes > 8)
    return fail(c, MGU_ERR_INVALID, "bad dice_loss args (1 <= num_classes <= 8)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int nb = std::min(128, nblocks(HW));
  double* ws;
  int rc = loss_scratch(c, (size_t)B * nb * 24, &ws);
  if (rc) return rc;
  if (!c->err_word) {
    HIPCHK(c, hipHostMalloc((void**)&c->err_word, sizeof(int), hipHostMallocMapped));
    *c->err_word = 0;
  }
  int* err_dev = nullptr;
  HIPCHK(c, hipHostGetDevicePointer((void**)&err_dev, c->err_word, 0));
  if (num_classes <= 4) {
    hipLaunchKernelGGL(dice_partial_kernel<4>, dim3(nb, B), dim3(256), 0, s, (const float*)logits_dev, labels_dev, HW, num_classes, ls_n,
                       ls_c, ls_p, ws, err_dev);
    hipLaunchKernelGGL(dice_final_kernel<4>, dim3(1), dim3(1), 0, s, ws, B, nb, num_classes, (double)smooth, loss_dev);
  } else {
    hipLaunchKernelGGL(dice_partial_kernel<8>, dim3(nb, B), dim3(256), 0, s, (const float*)logits_dev, labels_dev, HW, num_classes, ls_n,
                       ls_c, ls_p, ws, err_dev);
    hipLaunchKernelGGL(dice_final_kernel<8>, dim3(1), dim3(1), 0, s, ws, B, nb, num_classes, (double)smooth, loss_dev);
  }
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_tv_loss_backward(mgu_ctx* c, const void* x_dev, int B, int Cc, int H, int W, int64_t xs_n, int64_t xs_c, int64_t xs_h,
                         int64_t xs_w, float weight, float grad_scale, const float* grad_scale_dev, void* dx_dev, int64_t ds_n, int64_t ds_c,
                         int64_t ds_h, int64_t ds_w, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!x_dev || !dx_dev || B < 1 || Cc < 1 || H < 2 || W < 2) return fail(c, MGU_ERR_INVALID, "bad tv_loss_backward args (H, W >= 2)");
  HIPCHK(c, hipSetDevice(c->device));
  const int nb = nblocks((int64_t)B * Cc * H * W);
  // weight * (h_tv / count_h + w_tv / count_w) / B, each difference d contributing 2 d to its end points
  const float kh = (float)(2.0 * weight / ((double)(H - 1) * W) / B), kw = (float)(2.0 * weight / ((double)H * (W - 1)) / B);
  hipLaunchKernelGGL(tv_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)hip_stream, (const float*)x_dev, B, Cc, H, W, xs_n, xs_c, xs_h,
                     xs_w, kh, kw, grad_scale, grad_scale_dev, (float*)dx_dev, ds_n, ds_c, ds_h, ds_w);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_dice_loss_backward(mgu_ctx* c, const void* logits_dev, const int64_t* labels_dev, int B, int64_t HW, int num_classes,
                           int64_t ls_n, int64_t ls_c, int64_t ls_p, float smooth, float grad_scale, const float* grad_scale_dev,
                           void* dlogits_dev, int64_t ds_n, int64_t ds_c, int64_t ds_p, int accumulate, float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!logits_dev || !labels_dev || !dlogits_dev || B < 1 || HW < 1 || num_classes < 1 || num_classes > 8)
    return fail(c, MGU_ERR_INVALID, "bad dice_loss_backward args (1 <= num_classes <= 8)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int nb = std::min(128, nblocks(HW));
  double* ws;
  int rc = loss_scratch(c, (size_t)B * nb * 24 + (size_t)B * 8, &ws);   // partial records + the coefficient table (floats)
  if (rc) return rc;
  float* coef = (float*)(ws + (size_t)B * nb * 24);
  if (!c->err_word) {
    HIPCHK(c, hipHostMalloc((void**)&c->err_word, sizeof(int), hipHostMallocMapped));
    *c->err_word = 0;
  }
  int* err_dev = nullptr;
  HIPCHK(c, hipHostGetDevicePointer((void**)&err_dev, c->err_word, 0));
#define MGU_DICE_BWD(NC)                                                                                                             \
  hipLaunchKernelGGL(dice_partial_kernel<NC>, dim3(nb, B), dim3(256), 0, s, (const float*)logits_dev, labels_dev, HW, num_classes, ls_n, \
                     ls_c, ls_p, ws, err_dev);                                                                                       \
  if (loss_dev) hipLaunchKernelGGL(dice_final_kernel<NC>, dim3(1), dim3(1), 0, s, ws, B, nb, num_classes, (double)smooth, loss_dev);   \
  hipLaunchKernelGGL(dice_coef_kernel<NC>, dim3((B * NC + 63) / 64), dim3(64), 0, s, ws, B, nb, num_classes, (double)smooth, coef);    \
  hipLaunchKernelGGL(dice_bwd_kernel<NC>, dim3(nb, B), dim3(256), 0, s, (const float*)logits_dev, labels_dev, HW, num_classes, ls_n,   \
                     ls_c, ls_p, coef, grad_scale, grad_scale_dev, (float*)dlogits_dev, ds_n, ds_c, ds_p, accumulate)
  if (num_classes <= 4) {
    MGU_DICE_BWD(4);
  } else {
    MGU_DICE_BWD(8);
  }
#undef MGU_DICE_BWD
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_feature_consistency_loss_backward(mgu_ctx* c, const void* f_unet_dev, const void* f_graph_dev, const void* y_dev, int B, int N,
                                          int D, float margin, float grad_scale, const float* grad_scale_dev, void* d_f_unet_dev,
                                          void* d_f_graph_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!f_unet_dev || !f_graph_dev || !y_dev || (!d_f_unet_dev && !d_f_graph_dev) || B < 1 || N < 1 || D < 4 || (D & 3))
    return fail(c, MGU_ERR_INVALID, "bad feature_consistency_loss_backward args (D a multiple of 4)");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t rows = (int64_t)B * N;
  hipLaunchKernelGGL(featcons_bwd_kernel, dim3(nblocks(rows * 16)), dim3(256), 0, (hipStream_t)hip_stream, (const float*)f_unet_dev,
                     (const float*)f_graph_dev, (const float*)y_dev, rows, D, margin, grad_scale / (float)B, grad_scale_dev,
                     (float*)d_f_unet_dev, (float*)d_f_graph_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_loss_sync_check(mgu_ctx* c, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize((hipStream_t)hip_stream));
  if (c->err_word && *(volatile int*)c->err_word) {
    const int w = *(volatile int*)c->err_word;
    *(volatile int*)c->err_word = 0;
    return fail(c, MGU_ERR_INVALID, "%s", mgud::err_word_message(w));
  }
  return MGU_OK;
}

int mgu_feature_consistency_loss(mgu_ctx* c, const void* f_unet_dev, const void* f_graph_dev, const void* y_dev, int B, int N, int D,
                                 float margin, float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!f_unet_dev || !f_graph_dev || !y_dev || !loss_dev || B < 1 || N < 1 || D < 4 || (D & 3))
    return fail(c, MGU_ERR_INVALID, "bad feature_consistency_loss args (D a multiple of 4)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const int64_t rows = (int64_t)B * N;
  const int nb = nblocks(rows * 16);
  double* ws;
  int rc = loss_scratch(c, (size_t)nb, &ws);
  if (rc) return rc;
  hipLaunchKernelGGL(featcons_partial_kernel, dim3(nb), dim3(256), 0, s, (const float*)f_unet_dev, (const float*)f_graph_dev,
                     (const float*)y_dev, rows, D, margin, ws);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(1), 0, s, ws, nb, 1.0 / B, loss_dev);   // sum over patches, mean over the batch (:123)
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_elliptical_shape_loss_masks(mgu_ctx* c, const uint8_t* masks_dev, int num_objects, int H, int W, float epsilon, float* loss_dev,
                                    void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!loss_dev || num_objects < 0 || H < 1 || W < 1 || (num_objects > 0 && !masks_dev)) return fail(c, MGU_ERR_INVALID, "bad shape_loss args");
  HIPCHK(c, hipSetDevice(c->device));
  if (num_objects == 0) {
    HIPCHK(c, hipMemsetAsync(loss_dev, 0, sizeof(float), (hipStream_t)hip_stream));
    return MGU_OK;
  }
  return shape_loss<0>(c, masks_dev, num_objects, H, W, 0, 0, 0, 0, epsilon, loss_dev, (hipStream_t)hip_stream);
}

int mgu_elliptical_shape_loss_probs(mgu_ctx* c, const void* probs_dev, int B, int num_classes, int H, int W, int64_t ps_n, int64_t ps_c,
                                    int64_t ps_p, float epsilon, float* loss_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!probs_dev || !loss_dev || B < 1 || num_classes < 1 || H < 1 || W < 1) return fail(c, MGU_ERR_INVALID, "bad shape_loss args");
  HIPCHK(c, hipSetDevice(c->device));
  if (num_classes <= 1) {   // no foreground class to analyse (:63-64)
    HIPCHK(c, hipMemsetAsync(loss_dev, 0, sizeof(float), (hipStream_t)hip_stream));
    return MGU_OK;
  }
  return shape_loss<1>(c, probs_dev, B, H, W, num_classes, ps_n, ps_c, ps_p, epsilon, loss_dev, (hipStream_t)hip_stream);
}

}  // extern "C"
