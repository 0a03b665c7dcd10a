// Per-channel building blocks the DetectionHead needs besides the convolutions (second half;
// model/fusion_detection/detection_head.py:31-40, 93, 101-104):
//   * mgu_channel_affine_nhwc: y = act(scale[c] * x + shift[c]) over an (M, C) NHWC view -- the BatchNorm2d that FOLLOWS a
//     ReLU there (Conv -> ReLU -> BN, :33-38: the affine cannot be folded into the convolution in front of the ReLU nor,
//     because of the zero padding, into the one behind it) and the sigmoid of the box / confidence heads;
//   * mgu_channel_sum_nhwc: column sums of an (M, C) view -- AdaptiveAvgPool2d((1, 1)) (:39) is this sum per image
//     times 1/(H W), and the second BatchNorm, being affine, commutes with the mean.
// Both are single streaming passes (HBM-bound); the sum reuses the slotted double accumulators of the training
// reductions (train_kernels.hip).
#include "common.h"
#include "ctx.h"

namespace mgu {

typedef float f32x4c __attribute__((ext_vector_type(4)));

template <int ACT>
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int64_t M, int C, float* __restrict__ y,
                                                             int ldy) {
  const int Q = C >> 2;
  const int64_t total = M * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % Q);
    const int64_t m = i / Q;
    f32x4c v = *reinterpret_cast<const f32x4c*>(x + m * ldx + cq * 4);
    if (scale) v *= *reinterpret_cast<const f32x4c*>(scale + cq * 4);
    if (shift) v += *reinterpret_cast<const f32x4c*>(shift + cq * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ACT == 1) v[e] = fmaxf(v[e], 0.f);
      if (ACT == 2) v[e] = 1.f / (1.f + expf(-v[e]));
    }
    *reinterpret_cast<f32x4c*>(y + m * ldy + cq * 4) = v;
  }
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

extern "C" {

int mgu_channel_affine_nhwc(mgu_ctx* c, const float* x_dev, int ldx, int64_t M, int C, const float* scale_dev, const float* shift_dev,
                            int act, float* y_dev, int ldy, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (M < 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C || act < 0 || act > 2)
    return fail(c, MGU_ERR_INVALID, "mgu_channel_affine_nhwc: unsupported arguments (C, ldx, ldy multiples of 4; act in 0..2)");
  if (M == 0) return MGU_OK;
  if (!x_dev || !y_dev) return fail(c, MGU_ERR_INVALID, "mgu_channel_affine_nhwc: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  int64_t blocks = (M * (C >> 2) + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 g((unsigned)blocks), b(256);
  if (act == 0) hipLaunchKernelGGL(channel_affine_kernel<0>, g, b, 0, s, x_dev, ldx, scale_dev, shift_dev, M, C, y_dev, ldy);
  else if (act == 1) hipLaunchKernelGGL(channel_affine_kernel<1>, g, b, 0, s, x_dev, ldx, scale_dev, shift_dev, M, C, y_dev, ldy);
  else hipLaunchKernelGGL(channel_affine_kernel<2>, g, b, 0, s, x_dev, ldx, scale_dev, shift_dev, M, C, y_dev, ldy);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_channel_sum_nhwc(mgu_ctx* c, const float* x_dev, int ldx, int64_t M, int C, float* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (M < 1 || C < 4 || (C & 3) || C > 1024 || (ldx & 3) || ldx < C)
    return fail(c, MGU_ERR_INVALID, "mgu_channel_sum_nhwc: unsupported arguments (4 <= C <= 1024, C and ldx multiples of 4, M >= 1)");
  if (!x_dev || !out_dev) return fail(c, MGU_ERR_INVALID, "mgu_channel_sum_nhwc: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t need = chan_reduce_work_bytes(C < 64 ? 64 : C);
  if (c->redws_bytes < need) {   // the slots must be zero between reductions: a fresh allocation is cleared once
    int rc = ensure(c, &c->redws, &c->redws_bytes, need);
    if (rc) return rc;
    HIPCHK(c, hipMemset(c->redws, 0, need));
  }
  HIPCHK(c, launch_colsum(x_dev, ldx, M, C, (double*)c->redws, out_dev, (hipStream_t)hip_stream));
  return MGU_OK;
}

// One sum per image: x (B, M, C) with row pitch ldx, out (B, C).  Each image's reduction fills the chip on its own (a 512^2
// map is 2^18 rows), so the images are B back-to-back launches inside the library instead of B trips through the binding.
int mgu_channel_sum_images_nhwc(mgu_ctx* c, const float* x_dev, int ldx, int B, int64_t M, int C, float* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (B < 1) return fail(c, MGU_ERR_INVALID, "mgu_channel_sum_images_nhwc: B must be >= 1");
  for (int b = 0; b < B; ++b) {
    int rc = mgu_channel_sum_nhwc(c, x_dev ? x_dev + (size_t)b * M * ldx : nullptr, ldx, M, C, out_dev ? out_dev + (size_t)b * C : nullptr,
                                  hip_stream);
    if (rc) return rc;
  }
  return MGU_OK;
}

}  // extern "C"
