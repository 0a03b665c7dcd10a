// Region stage + feature fusion of the e2e forward (first half):
//   * label-mean pooling of the GAT-refined patch features into K region nodes per image
//     (scripts/train_end_to_end.py:366-373; an empty segment keeps a zero feature);
//   * after the region GAT (mgu_gat_layer_forward on the K-node graphs, :382-390): region embedding -> its patches
//     (:403-406) -> patch grid -> pixels by nearest interpolation (:410-421) -> channel concat with the U-Net feature
//     (FeatureFusion, model/fusion_detection/feature_fusion.py:78,145-150), in ONE pass that writes the fused NHWC
//     tensor: the (B, D, H, W) pixel-mapped F_g of the reference (537 MB at batch 8, written, re-read by torch.stack and
//     again by torch.cat) is never materialised.
// Both kernels are pure data movement: HBM-bound, 16-byte lanes over channels.
#include "ctx.h"

namespace mgu {

typedef float f32x4r __attribute__((ext_vector_type(4)));

// one workgroup per (image, segment); thread = channel quad x patch lane
__global__ __launch_bounds__(256) void region_pool_kernel(const float* __restrict__ feats, const int32_t* __restrict__ hard, int Np, int D,
                                                          int K, float* __restrict__ out) {
  extern __shared__ float red[];   // [lanes][D]
  const int b = blockIdx.x / K, k = blockIdx.x - b * K;
  const int q = D >> 2, cq = threadIdx.x % q, pl = threadIdx.x / q, npl = 256 / q;
  f32x4r acc = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  if (pl < npl) {
    // four patches per trip, labels and rows loaded unconditionally (independent addresses: one round trip per trip)
    for (int p0 = pl; p0 < Np; p0 += 4 * npl) {
      int h[4];
      f32x4r f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = min(p0 + u * npl, Np - 1);
        h[u] = hard[(size_t)b * Np + p];
        f[u] = *reinterpret_cast<const f32x4r*>(feats + ((size_t)b * Np + p) * D + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = (p0 + u * npl < Np) && h[u] == k;
        acc += in ? f[u] : f32x4r{0.f, 0.f, 0.f, 0.f};
        cnt += in ? 1 : 0;
      }
    }
  }
  __shared__ int cnts[256];
  cnts[threadIdx.x] = (cq == 0 && pl < npl) ? cnt : 0;
  if (pl < npl) *reinterpret_cast<f32x4r*>(red + (size_t)pl * D + cq * 4) = acc;
  __syncthreads();
  if (threadIdx.x < q) {
    f32x4r s = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    for (int l = 0; l < npl; ++l) {
      s += *reinterpret_cast<const f32x4r*>(red + (size_t)l * D + threadIdx.x * 4);
      n += cnts[l * q];
    }
    const float inv = n > 0 ? 1.f / (float)n : 0.f;   // mask_k.sum() > 0 (:371-372)
    *reinterpret_cast<f32x4r*>(out + ((size_t)b * K + k) * D + threadIdx.x * 4) = s * inv;
  }
}

// thread = (pixel, channel quad of the fused pixel)
__global__ __launch_bounds__(256) void region_fuse_kernel(const float* __restrict__ fu, int Cu, const float* __restrict__ emb,
                                                          const int32_t* __restrict__ hard, int B, int H, int W, int nph, int npw,
                                                          int K, int D, float sy, float sx, float* __restrict__ out) {
  const int Q = (Cu + D) >> 2;
  const int64_t total = (int64_t)B * H * W * Q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % Q);
    const int64_t pix = i / Q;
    f32x4r v;
    if (cq * 4 < Cu) {
      v = *reinterpret_cast<const f32x4r*>(fu + pix * Cu + cq * 4);
    } else {
      const int x = (int)(pix % W);
      const int64_t r = pix / W;
      const int y = (int)(r % H), b = (int)(r / H);
      // torch 'nearest': src = min(floor(dst * (in / out)), in - 1), the scale formed in fp32 (:416-420)
      const int py = min((int)floorf((float)y * sy), nph - 1), px = min((int)floorf((float)x * sx), npw - 1);
      const int lbl = hard[(size_t)b * nph * npw + py * npw + px];
      v = *reinterpret_cast<const f32x4r*>(emb + ((size_t)b * K + lbl) * D + (cq * 4 - Cu));
    }
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4r*>(out + pix * (Cu + D) + cq * 4));   // write-once stream: keep it out of L2
  }
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

extern "C" {

int mgu_region_mean_pool(mgu_ctx* c, const float* feats_dev, const int32_t* hard_dev, int B, int Np, int D, int K, float* out_dev,
                         void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (B <= 0 || Np <= 0 || K <= 0 || D <= 0 || (D & 3) || D > 1024)
    return fail(c, MGU_ERR_INVALID, "mgu_region_mean_pool: unsupported sizes B=%d Np=%d D=%d K=%d (D %% 4 == 0, D <= 1024)", B, Np, D, K);
  if (!feats_dev || !hard_dev || !out_dev) return fail(c, MGU_ERR_INVALID, "mgu_region_mean_pool: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  const int q = D >> 2, npl = 256 / q;
  if (npl < 1) return fail(c, MGU_ERR_INVALID, "mgu_region_mean_pool: D too wide");
  hipLaunchKernelGGL(region_pool_kernel, dim3(B * K), dim3(256), (size_t)npl * D * sizeof(float), (hipStream_t)hip_stream, feats_dev, hard_dev,
                     Np, D, K, out_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_region_fuse_nhwc(mgu_ctx* c, const float* fu_nhwc_dev, int Cu, const float* region_emb_dev, const int32_t* hard_dev, int B, int H,
                         int W, int nph, int npw, int K, int D, float* out_nhwc_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (B <= 0 || H <= 0 || W <= 0 || nph <= 0 || npw <= 0 || K <= 0 || D <= 0 || (D & 3) || Cu < 0 || (Cu & 3))
    return fail(c, MGU_ERR_INVALID, "mgu_region_fuse_nhwc: unsupported sizes (channel counts must be multiples of 4)");
  if ((Cu > 0 && !fu_nhwc_dev) || !region_emb_dev || !hard_dev || !out_nhwc_dev) return fail(c, MGU_ERR_INVALID, "mgu_region_fuse_nhwc: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t total = (int64_t)B * H * W * ((Cu + D) >> 2);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(region_fuse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, fu_nhwc_dev, Cu, region_emb_dev, hard_dev,
                     B, H, W, nph, npw, K, D, (float)nph / (float)H, (float)npw / (float)W, out_nhwc_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

}  // extern "C"
