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
struct IgemmDesc {
  const float* in;
  const float* w;
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
};

hipError_t launch_igemm_f32(const IgemmDesc& d, hipStream_t s);
void set_use_halo(bool on);

// elementwise.hip
hipError_t launch_pack_input(const float* x, float* out, int B, int C, int Cp, int H, int W, int64_t sn, int64_t sc,
                             int64_t sh, int64_t sw, hipStream_t s);
hipError_t launch_maxpool2(const float* in, int ldin, float* out, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_patch_mean(const float* feat, float* out, int B, int H, int W, int C, int patch, hipStream_t s);
hipError_t launch_pack_conv_w(const float* w_oihw, float* wp, int Cout, int Cin, int Cp, int KS, int Kp, hipStream_t s);
hipError_t launch_pack_convt_w(const float* w_iohw, float* wp, int Cin, int Cout, int Kp, hipStream_t s);
hipError_t launch_bn_fold(const float* bias, const float* gamma, const float* beta, const float* mean, const float* var,
                          float eps, float* scale, float* shift, int C, hipStream_t s);
hipError_t launch_bias_tile(const float* bias, float* shift, int C, int reps, hipStream_t s);
hipError_t launch_argmax(const float* logits, int64_t npix, int C, int64_t* pred, hipStream_t s);

// gat.hip
hipError_t launch_gat_st(const float* Wh, const float* a, float* st, int N, int heads, int Fh, hipStream_t s);
hipError_t launch_gat_edge_max(const float* st, const int32_t* rowptr, const int32_t* col, const int32_t* graph_ptr,
                               int num_graphs, int N, int heads, float alpha, unsigned* gmax_enc, hipStream_t s);
hipError_t launch_gat_aggregate(const float* Wh, const float* st, const int32_t* rowptr, const int32_t* col,
                                const int32_t* graph_ptr, int num_graphs, const unsigned* gmax_enc, int N, int heads,
                                int Fh, int concat, float alpha, float* out, hipStream_t s);

}  // namespace mgu
