// Input / output pipeline around the network on the device and the bilinear resize FeatureFusion needs:
//   mgu_resize_bilinear_nhwc    F.interpolate(mode='bilinear', align_corners=False)   model/fusion_detection/feature_fusion.py:69-76, 140-144
//   mgu_preprocess_image_u8     ImagePreprocessor.preprocess                          preprocessing/image_preprocessing/image_preprocess.py:26-31, 57-85
//   mgu_preprocess_mask_u8      ImagePreprocessor.preprocess_mask                     image_preprocess.py:87-126
//   mgu_sobel_edges_u8          EdgeDetector.sobel_edges                              preprocessing/graph_feature_processing/edge_detection.py:14-44
//   mgu_equalize_hist_rgb_u8    HistogramEqualizer.equalize_histogram_rgb             preprocessing/graph_feature_processing/histogram_equalization.py:13-35
//   mgu_patch_mean_u8           image_to_patches(...).mean(dim=[1,2,3]) / [2,3]       scripts/graph_refinement.py:97-104
//   mgu_colorize_labels         postprocess_segmentation                              scripts/infer_segmentation.py:20-51
// The reference does these on the HOST with cv2 / PIL (torchvision.transforms.Resize on a PIL image = PIL's antialiased
// BILINEAR resample in 8-bit fixed point).  Byte and integer work: every kernel reproduces the library arithmetic exactly --
// the same 22-bit coefficient tables and rounding as PIL's ImagingResample, OpenCV's 14-bit colour-conversion constants,
// reflect-101 borders, double-precision normalisation -- so the outputs are comparable bit for bit.  All of it is HBM bound.
#include <math.h>

#include <algorithm>
#include <vector>

#include "ctx.h"

namespace mgu {

// ---- F.interpolate bilinear, align_corners = False (aten upsample_bilinear2d): src = max(0, scale (dst + 0.5) - 0.5) -------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int ld_in, int B, int Hi, int Wi, int Cc,
                                                              float* __restrict__ out, int ld_out, int c_off, int Ho, int Wo,
                                                              float sy, float sx) {
  const int Q = Cc >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * Q;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % Q);
    int64_t r = i / Q;
    const int x = (int)(r % Wo);
    r /= Wo;
    const int y = (int)(r % Ho), n = (int)(r / Ho);
    const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* base = in + (size_t)n * Hi * Wi * ld_in + cq * 4;
    const float4 a = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Wi + x0) * ld_in);
    const float4 b = *reinterpret_cast<const float4*>(base + ((size_t)y0 * Wi + x1) * ld_in);
    const float4 c = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Wi + x0) * ld_in);
    const float4 d = *reinterpret_cast<const float4*>(base + ((size_t)y1 * Wi + x1) * ld_in);
    float4 o;   // aten: h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (w0lambda * c + w1lambda * d)
    o.x = hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
    o.y = hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
    o.z = hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
    o.w = hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
    *reinterpret_cast<float4*>(out + (((size_t)n * Ho + y) * Wo + x) * ld_out + c_off + cq * 4) = o;
  }
}

// ---- PIL ImagingResample, 8 bits per channel, BILINEAR (triangle) filter with antialiasing ------------------------------------------
constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;
struct ResampleTable {
  int ksize = 0;
  std::vector<int> bounds;   // [out][2]: first source index, tap count
  std::vector<int> kk;       // [out][ksize] fixed-point weights
};
static ResampleTable pil_bilinear_coeffs(int inSize, int outSize) {
  ResampleTable t;
  const double scale = (double)inSize / outSize;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;        // BILINEAR: support 1.0
  t.ksize = (int)ceil(support) * 2 + 1;
  t.bounds.assign((size_t)outSize * 2, 0);
  t.kk.assign((size_t)outSize * t.ksize, 0);
  std::vector<double> k(t.ksize);
  for (int xx = 0; xx < outSize; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      const double w = v < 1.0 ? 1.0 - v : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      t.kk[(size_t)xx * t.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PIL_PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PIL_PRECISION_BITS));
    }
    t.bounds[2 * xx] = xmin, t.bounds[2 * xx + 1] = xmax;
  }
  return t;
}
__device__ __forceinline__ uint8_t pil_clip8(int v) {
  v >>= PIL_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// one pass: out[o][p][c] = clip8(half + sum_k in[bounds[o].first + k][p][c] * kk[o][k]); `o` runs along the resampled axis with
// element stride s_o, `p` along the other axis with stride s_p (elements of C channels)
__global__ __launch_bounds__(256) void pil_resample_pass_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int nOut, int nOther,
                                                                int Cc, int64_t in_so, int64_t in_sp, int64_t out_so, int64_t out_sp,
                                                                const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const int64_t total = (int64_t)nOut * nOther * Cc;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cc);
    int64_t r = i / Cc;
    const int p = (int)(r % nOther), o = (int)(r / nOther);
    const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
    int ss = 1 << (PIL_PRECISION_BITS - 1);
    for (int k = 0; k < cnt; ++k) ss += (int)in[(first + k) * in_so + p * in_sp + c] * kk[o * ksize + k];
    out[o * out_so + p * out_sp + c] = pil_clip8(ss);
  }
}
// ToTensor + Normalize (image_preprocess.py:29-30): v = u8 / 255 (fp32), (v - mean) / std, written CHW; bgr = 1 swaps channels 0 and 2
__global__ __launch_bounds__(256) void to_tensor_normalize_kernel(const uint8_t* __restrict__ in, int H, int W, int Cin, int bgr,
                                                                  float m0, float m1, float m2, float s0, float s1, float s2,
                                                                  float* __restrict__ out, int64_t os_c, int64_t os_h, int64_t os_w) {
  const int64_t total = (int64_t)H * W * 3;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const int64_t pix = i / 3;
    const int x = (int)(pix % W), y = (int)(pix / W);
    const int sc = Cin == 1 ? 0 : (bgr ? 2 - c : c);       // grey -> three equal channels (cv2.COLOR_GRAY2RGB, :79-80)
    const float v = (float)in[pix * Cin + sc] / 255.f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[c * os_c + y * os_h + x * os_w] = (v - mean) / sd;
  }
}

// ---- cv2.resize(INTER_NEAREST) + np.clip + long (image_preprocess.py:117-125) --------------------------------------------------------
__global__ __launch_bounds__(256) void mask_nearest_kernel(const uint8_t* __restrict__ in, int Hs, int Ws, int64_t* __restrict__ out, int H, int W,
                                                           int num_classes) {
  const double ify = 1.0 / ((double)H / (double)Hs), ifx = 1.0 / ((double)W / (double)Ws);   // cv::resize: inv_scale = dsize / ssize
  const int64_t total = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)(i / W);
    const int sy = min((int)floor(y * ify), Hs - 1), sx = min((int)floor(x * ifx), Ws - 1);
    const int v = in[(size_t)sy * Ws + sx];
    out[i] = (int64_t)min(max(v, 0), num_classes - 1);
  }
}

// ---- Sobel edge magnitude (edge_detection.py:28-44): RGB2GRAY (14-bit fixed point), 3x3 Sobel with reflect-101 borders in exact
// integers, sqrt(gx^2 + gy^2) / max * 255 in double, truncated to uint8 -------------------------------------------------------------------
// cv2.COLOR_RGB2GRAY on 8-bit data: OpenCV 3.4 / 4.x use 15-bit coefficients (RY15 9798, GY15 19235, BY15 3735, gray_shift 15); only
// the YUV / YCrCb conversions below keep the 14-bit ones (yuv_shift 14)
__device__ __forceinline__ int cv_gray(const uint8_t* p) { return (p[0] * 9798 + p[1] * 19235 + p[2] * 3735 + (1 << 14)) >> 15; }
__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__global__ __launch_bounds__(256) void sobel_mag2_kernel(const uint8_t* __restrict__ rgb, int H, int W, int* __restrict__ mag2,
                                                         unsigned* __restrict__ max2) {
  unsigned local = 0;
  const int64_t total = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)(i / W);
    int g[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = H > 1 ? reflect101(y + dy - 1, H) : 0, xx = W > 1 ? reflect101(x + dx - 1, W) : 0;
        g[dy][dx] = cv_gray(rgb + ((size_t)yy * W + xx) * 3);
      }
    const int gx = (g[0][2] + 2 * g[1][2] + g[2][2]) - (g[0][0] + 2 * g[1][0] + g[2][0]);
    const int gy = (g[2][0] + 2 * g[2][1] + g[2][2]) - (g[0][0] + 2 * g[0][1] + g[0][2]);
    const int m = gx * gx + gy * gy;
    mag2[i] = m;
    local = max(local, (unsigned)m);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local = max(local, (unsigned)__shfl_xor((int)local, off));
  if ((threadIdx.x & 63) == 0 && local) atomicMax(max2, local);
}
__global__ __launch_bounds__(256) void sobel_norm_kernel(const int* __restrict__ mag2, const unsigned* __restrict__ max2, int64_t n,
                                                         uint8_t* __restrict__ out) {
  const double mx = sqrt((double)*max2);
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = mx > 0.0 ? (uint8_t)(sqrt((double)mag2[i]) / mx * 255.0) : (uint8_t)0;   // (e / max * 255).astype(uint8): truncation
}

// ---- histogram equalisation of the luminance (histogram_equalization.py:27-35): cv2 RGB2YUV / equalizeHist / YUV2RGB ----------------
__device__ __forceinline__ int cv_descale14(int v) { return (v + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ void cv_rgb2yuv(const uint8_t* p, int& Y, int& U, int& V) {
  Y = cv_descale14(p[0] * 4899 + p[1] * 9617 + p[2] * 1868);
  U = sat8(cv_descale14((p[2] - Y) * 8061 + (128 << 14)));    // B2UF = 0.492
  V = sat8(cv_descale14((p[0] - Y) * 14369 + (128 << 14)));   // R2VF = 0.877
}
__global__ __launch_bounds__(256) void yuv_hist_kernel(const uint8_t* __restrict__ rgb, int64_t n, unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int Y, U, V;
    cv_rgb2yuv(rgb + i * 3, Y, U, V);
    atomicAdd(&h[Y], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void equalize_lut_kernel(const unsigned* __restrict__ hist, int64_t total, uint8_t* __restrict__ lut) {
  // cv::equalizeHist: first non-empty bin i0; scale = 255 / (total - hist[i0]); lut[i] = saturate(round(cumsum_{i0 < j <= i} * scale))
  if (threadIdx.x != 0) return;
  int i0 = 0;
  while (i0 < 256 && !hist[i0]) ++i0;
  if (i0 == 256 || hist[i0] == total) {
    for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)(i0 < 256 ? i0 : i);   // a constant image maps to itself
    return;
  }
  const float scale = 255.f / (float)(total - hist[i0]);
  int sum = 0;
  for (int i = 0; i < 256; ++i) {
    if (i <= i0) {
      lut[i] = 0;
      continue;
    }
    sum += hist[i];
    lut[i] = (uint8_t)sat8((int)rintf((float)sum * scale));
  }
}
__global__ __launch_bounds__(256) void equalize_apply_kernel(const uint8_t* __restrict__ rgb, int64_t n, const uint8_t* __restrict__ lut,
                                                             uint8_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int Y, U, V;
    cv_rgb2yuv(rgb + i * 3, Y, U, V);
    Y = lut[Y];
    const int u = U - 128, v = V - 128;
    out[i * 3 + 0] = (uint8_t)sat8(Y + cv_descale14(v * 18678));                  // V2RI = 1.140
    out[i * 3 + 1] = (uint8_t)sat8(Y + cv_descale14(u * -6472 + v * -9519));      // U2GI = -0.395, V2GI = -0.581
    out[i * 3 + 2] = (uint8_t)sat8(Y + cv_descale14(u * 33292));                  // U2BI = 2.032
  }
}

// ---- per-patch mean of a uint8 map, zero padded bottom / right as image_to_patches (graph_refinement.py:97-104) ----------------------
// per_channel = 0: one value per patch over all channels (sobel: mean(dim=[1,2,3]) of a 1-channel map; histeq as written in the script)
__global__ __launch_bounds__(256) void patch_mean_u8_kernel(const uint8_t* __restrict__ img, int H, int W, int Cc, int patch, int npw, int per_channel,
                                                            float* __restrict__ out) {
  const int pidx = blockIdx.x, py = pidx / npw, px = pidx - py * npw;
  __shared__ unsigned acc[4];
  if (threadIdx.x < 4) acc[threadIdx.x] = 0;
  __syncthreads();
  unsigned s[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < patch * patch; i += 256) {
    const int y = py * patch + i / patch, x = px * patch + i % patch;
    if (y < H && x < W)
      for (int c = 0; c < Cc; ++c) s[per_channel ? c : 0] += img[((size_t)y * W + x) * Cc + c];
  }
  for (int c = 0; c < 4; ++c)
    if (s[c]) atomicAdd(&acc[c], s[c]);
  __syncthreads();
  const int nout = per_channel ? Cc : 1;
  if ((int)threadIdx.x < nout)
    out[(size_t)pidx * nout + threadIdx.x] = (float)((double)acc[threadIdx.x] / ((double)patch * patch * (per_channel ? 1 : Cc)));
}

// ---- labels -> colour map (infer_segmentation.py:36-49) and uint8 label map (:123) -----------------------------------------------------
__global__ __launch_bounds__(256) void colorize_kernel(const int64_t* __restrict__ labels, int64_t n, const uint8_t* __restrict__ palette,
                                                       int num_classes, uint8_t* __restrict__ vis, uint8_t* __restrict__ lab8) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t l = labels[i];
    const bool ok = l >= 0 && l < num_classes;          // pixels of any other label stay black, as the reference's zero-initialised map
    vis[i * 3 + 0] = ok ? palette[l * 3 + 0] : 0;
    vis[i * 3 + 1] = ok ? palette[l * 3 + 1] : 0;
    vis[i * 3 + 2] = ok ? palette[l * 3 + 2] : 0;
    if (lab8) lab8[i] = (uint8_t)l;
  }
}

// ---- FeatureFusion's per-region branch (feature_fusion.py:84-138): out[pixel][c_off + d] = f_g[id[pixel]][d], zeros for an id outside
// [0, R) (the reference leaves those pixels of its zero-initialised map untouched) ------------------------------------------------------
__global__ __launch_bounds__(256) void region_map_gather_kernel(const float* __restrict__ table, int R, int D, const int64_t* __restrict__ ids,
                                                                int64_t npix, float* __restrict__ out, int ld_out, int c_off) {
  const int Q = D >> 2;
  const int64_t total = npix * Q;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % Q);
    const int64_t p = i / Q, id = ids[p];
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (id >= 0 && id < R) v = *reinterpret_cast<const float4*>(table + id * D + 4 * q);
    *reinterpret_cast<float4*>(out + p * ld_out + c_off + 4 * q) = v;
  }
}

}  // namespace mgu

using namespace mgu;
using namespace mgud;

namespace {
inline int nb(int64_t work) { return (int)std::max<int64_t>(1, std::min<int64_t>(256 * 8, (work + 255) / 256)); }
}

extern "C" {

int mgu_resize_bilinear_nhwc(mgu_ctx* c, const void* in_dev, int ld_in, int B, int Hi, int Wi, int Cc, void* out_dev, int ld_out, int c_off,
                             int Ho, int Wo, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!in_dev || !out_dev || B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || Cc < 4 || (Cc & 3) || (ld_in & 3) || (ld_out & 3) || (c_off & 3) ||
      ld_in < Cc || ld_out < c_off + Cc)
    return fail(c, MGU_ERR_INVALID, "bad resize_bilinear args (C, ld_in, ld_out, c_off multiples of 4)");
  HIPCHK(c, hipSetDevice(c->device));
  // aten area_pixel_compute_scale: scale = in / out in the accumulation type (float)
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(nb((int64_t)B * Ho * Wo * (Cc >> 2))), dim3(256), 0, (hipStream_t)hip_stream,
                     (const float*)in_dev, ld_in, B, Hi, Wi, Cc, (float*)out_dev, ld_out, c_off, Ho, Wo, (float)Hi / (float)Ho,
                     (float)Wi / (float)Wo);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_preprocess_image_u8(mgu_ctx* c, const uint8_t* img_dev, int Hs, int Ws, int channels, int bgr, int H, int W, const float* mean3,
                            const float* std3, void* out_dev, int64_t os_c, int64_t os_h, int64_t os_w, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!img_dev || !out_dev || !mean3 || !std3 || Hs < 1 || Ws < 1 || H < 1 || W < 1 || (channels != 1 && channels != 3))
    return fail(c, MGU_ERR_INVALID, "bad preprocess_image args (1 or 3 channels)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const bool need_h = W != Ws, need_v = H != Hs;
  const ResampleTable th = need_h ? pil_bilinear_coeffs(Ws, W) : ResampleTable(), tv = need_v ? pil_bilinear_coeffs(Hs, H) : ResampleTable();
  // scratch: [tables (int)] [horizontal result Hs x W x C] [final H x W x C]
  const size_t n_tab = th.bounds.size() + th.kk.size() + tv.bounds.size() + tv.kk.size();
  const size_t o_tmp = (n_tab * sizeof(int) + 255) / 256 * 256, o_fin = o_tmp + ((size_t)Hs * W * channels + 255) / 256 * 256;
  int rc = ensure(c, &c->imgws, &c->imgws_bytes, o_fin + (size_t)H * W * channels + 256);
  if (rc) return rc;
  char* ws = (char*)c->imgws;
  int* tab = (int*)ws;
  std::vector<int> host;
  host.reserve(n_tab);
  host.insert(host.end(), th.bounds.begin(), th.bounds.end());
  host.insert(host.end(), th.kk.begin(), th.kk.end());
  host.insert(host.end(), tv.bounds.begin(), tv.bounds.end());
  host.insert(host.end(), tv.kk.begin(), tv.kk.end());
  if (n_tab) {
    HIPCHK(c, hipMemcpyAsync(tab, host.data(), n_tab * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));   // `host` is a stack vector: the copy must have left it (tables are tiny; once per image)
  }
  const uint8_t* cur = img_dev;
  int curW = Ws;
  if (need_h) {   // horizontal pass first, as ImagingResample
    uint8_t* tmp = (uint8_t*)(ws + o_tmp);
    hipLaunchKernelGGL(pil_resample_pass_kernel, dim3(nb((int64_t)W * Hs * channels)), dim3(256), 0, s, cur, tmp, W, Hs, channels,
                       (int64_t)channels, (int64_t)Ws * channels, (int64_t)channels, (int64_t)W * channels, tab, tab + th.bounds.size(), th.ksize);
    cur = tmp, curW = W;
  }
  if (need_v) {
    uint8_t* fin = (uint8_t*)(ws + o_fin);
    const int* tb = tab + th.bounds.size() + th.kk.size();
    hipLaunchKernelGGL(pil_resample_pass_kernel, dim3(nb((int64_t)H * curW * channels)), dim3(256), 0, s, cur, fin, H, curW, channels,
                       (int64_t)curW * channels, (int64_t)channels, (int64_t)curW * channels, (int64_t)channels, tb, tb + tv.bounds.size(), tv.ksize);
    cur = fin;
  }
  hipLaunchKernelGGL(to_tensor_normalize_kernel, dim3(nb((int64_t)H * W * 3)), dim3(256), 0, s, cur, H, W, channels, bgr, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], (float*)out_dev, os_c, os_h, os_w);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_preprocess_mask_u8(mgu_ctx* c, const uint8_t* mask_dev, int Hs, int Ws, int H, int W, int num_classes, int64_t* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!mask_dev || !out_dev || Hs < 1 || Ws < 1 || H < 1 || W < 1 || num_classes < 1) return fail(c, MGU_ERR_INVALID, "bad preprocess_mask args");
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(mask_nearest_kernel, dim3(nb((int64_t)H * W)), dim3(256), 0, (hipStream_t)hip_stream, mask_dev, Hs, Ws, out_dev, H, W,
                     num_classes);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_sobel_edges_u8(mgu_ctx* c, const uint8_t* rgb_dev, int H, int W, uint8_t* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!rgb_dev || !out_dev || H < 1 || W < 1) return fail(c, MGU_ERR_INVALID, "bad sobel args");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  int rc = ensure(c, &c->imgws, &c->imgws_bytes, (size_t)H * W * sizeof(int) + 256);
  if (rc) return rc;
  unsigned* max2 = (unsigned*)c->imgws;
  int* mag2 = (int*)((char*)c->imgws + 256);
  HIPCHK(c, hipMemsetAsync(max2, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(sobel_mag2_kernel, dim3(nb((int64_t)H * W)), dim3(256), 0, s, rgb_dev, H, W, mag2, max2);
  hipLaunchKernelGGL(sobel_norm_kernel, dim3(nb((int64_t)H * W)), dim3(256), 0, s, mag2, max2, (int64_t)H * W, out_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_equalize_hist_rgb_u8(mgu_ctx* c, const uint8_t* rgb_dev, int H, int W, uint8_t* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!rgb_dev || !out_dev || H < 1 || W < 1) return fail(c, MGU_ERR_INVALID, "bad equalize_hist args");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)hip_stream;
  int rc = ensure(c, &c->imgws, &c->imgws_bytes, 256 * sizeof(unsigned) + 256);
  if (rc) return rc;
  unsigned* hist = (unsigned*)c->imgws;
  uint8_t* lut = (uint8_t*)(hist + 256);
  const int64_t n = (int64_t)H * W;
  HIPCHK(c, hipMemsetAsync(hist, 0, 256 * sizeof(unsigned), s));
  hipLaunchKernelGGL(yuv_hist_kernel, dim3(std::min(1024, nb(n))), dim3(256), 0, s, rgb_dev, n, hist);
  hipLaunchKernelGGL(equalize_lut_kernel, dim3(1), dim3(64), 0, s, hist, n, lut);
  hipLaunchKernelGGL(equalize_apply_kernel, dim3(nb(n)), dim3(256), 0, s, rgb_dev, n, lut, out_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_patch_mean_u8(mgu_ctx* c, const uint8_t* img_dev, int H, int W, int channels, int patch, int per_channel, float* out_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!img_dev || !out_dev || H < 1 || W < 1 || channels < 1 || channels > 4 || patch < 1 || patch > 4096)
    return fail(c, MGU_ERR_INVALID, "bad patch_mean_u8 args (1..4 channels)");
  HIPCHK(c, hipSetDevice(c->device));
  const int nph = (H + patch - 1) / patch, npw = (W + patch - 1) / patch;
  hipLaunchKernelGGL(patch_mean_u8_kernel, dim3(nph * npw), dim3(256), 0, (hipStream_t)hip_stream, img_dev, H, W, channels, patch, npw,
                     per_channel ? 1 : 0, out_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_region_map_gather_nhwc(mgu_ctx* c, const float* table_dev, int R, int D, const int64_t* ids_dev, int64_t npix, float* out_dev, int ld_out,
                               int c_off, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!table_dev || !ids_dev || !out_dev || R < 0 || D < 4 || (D & 3) || npix < 0 || (ld_out & 3) || (c_off & 3) || ld_out < c_off + D)
    return fail(c, MGU_ERR_INVALID, "bad region_map_gather args (D, ld_out, c_off multiples of 4)");
  HIPCHK(c, hipSetDevice(c->device));
  if (npix == 0) return MGU_OK;
  hipLaunchKernelGGL(region_map_gather_kernel, dim3(nb(npix * (D >> 2))), dim3(256), 0, (hipStream_t)hip_stream, table_dev, R, D, ids_dev, npix,
                     out_dev, ld_out, c_off);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

int mgu_colorize_labels(mgu_ctx* c, const int64_t* labels_dev, int64_t npix, const uint8_t* palette_dev, int num_classes, uint8_t* vis_dev,
                        uint8_t* labels_u8_dev, void* hip_stream) {
  if (!c) return MGU_ERR_INVALID;
  if (!labels_dev || !palette_dev || !vis_dev || npix < 0 || num_classes < 1) return fail(c, MGU_ERR_INVALID, "bad colorize args");
  HIPCHK(c, hipSetDevice(c->device));
  if (npix == 0) return MGU_OK;
  hipLaunchKernelGGL(colorize_kernel, dim3(nb(npix)), dim3(256), 0, (hipStream_t)hip_stream, labels_dev, npix, palette_dev, num_classes, vis_dev,
                     labels_u8_dev);
  HIPCHK(c, hipGetLastError());
  return MGU_OK;
}

}  // extern "C"
