// Micro-benchmark: cost of the exact three-way bf16 split (split3_pack of wino_f32.hip) with and without the six
// bf16 MFMAs it feeds, per wave and with one or two waves per SIMD.  Prints cycles (at the measured clock) per step.
// Build: hipcc -O3 --offload-arch=gfx950 -o split_cost split_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_pack(const float a, const float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
// round-to-nearest pieces via v_cvt_pk_bf16_f32 + remainders via v_dot2c_f32_bf16 (x3.h split3_pack_d): 7 VALU per pair instead of 11
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pack_d(const bf16x2 lo, const bf16x2 hi, const float a, const float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const bf16x2 q0 = {(__bf16)a, (__bf16)b};
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(q0, lo, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(q0, hi, b, false);
  const bf16x2 q1 = {(__bf16)ra, (__bf16)rb};
  const float sa = __builtin_amdgcn_fdot2_f32_bf16(q1, lo, ra, false), sb = __builtin_amdgcn_fdot2_f32_bf16(q1, hi, rb, false);
  const bf16x2 q2 = {(__bf16)sa, (__bf16)sb};
  p0 = __builtin_bit_cast(unsigned, q0), p1 = __builtin_bit_cast(unsigned, q1), p2 = __builtin_bit_cast(unsigned, q2);
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// WHAT: 1 = split only, 2 = MFMA only, 3 = both (software pipelined: MFMAs of step s with the split of step s+1)
template <int WHAT, int INTERLEAVE, int NCH, int DOT>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed, long long* cyc) {
  unsigned klo = 0x0000BF80u, khi = 0xBF800000u;
  asm volatile("" : "+v"(klo), "+v"(khi));
  const bf16x2 lo = __builtin_bit_cast(bf16x2, klo), hi = __builtin_bit_cast(bf16x2, khi);
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 v[2] = {f32x4{seed, seed * 3, seed * 5, seed * 7} + (float)threadIdx.x, f32x4{seed * 1.5f, seed * 2.5f, seed * 3.5f, seed * 4.5f}};
  u32x4 b[3];
  for (int i = 0; i < 3; ++i) b[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  u32x4 pc[2][3];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 3; ++i) pc[s][i] = b[i];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      if (WHAT & 2) {
        f32x16 t = acc[st], u = acc[st + 4];
        t = mfma_bf16(pc[st & 1][2], b[0], t);
        if (NCH == 2) u = mfma_bf16(pc[st & 1][2], b[1], u);
        t = mfma_bf16(pc[st & 1][0], b[2], t);
        if (NCH == 2) u = mfma_bf16(pc[st & 1][0], b[0], u);
        t = mfma_bf16(pc[st & 1][1], b[1], t);
        if (NCH == 2) u = mfma_bf16(pc[st & 1][1], b[2], u);
        t = mfma_bf16(pc[st & 1][1], b[0], t);
        if (NCH == 2) u = mfma_bf16(pc[st & 1][1], b[1], u);
        t = mfma_bf16(pc[st & 1][0], b[1], t);
        if (NCH == 2) u = mfma_bf16(pc[st & 1][0], b[2], u);
        t = mfma_bf16(pc[st & 1][0], b[0], t);
        if (NCH == 2) u = mfma_bf16(pc[st & 1][0], b[1], u);
        acc[st] = t;
        acc[st + 4] = u;
      }
      if (WHAT & 1) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            unsigned p0, p1, p2;
            if (DOT) split3_pack_d(lo, hi, v[hf][2 * e], v[hf][2 * e + 1], p0, p1, p2);
            else split3_pack(v[hf][2 * e], v[hf][2 * e + 1], p0, p1, p2);
            pc[(st + 1) & 1][0][hf * 2 + e] = p0, pc[(st + 1) & 1][1][hf * 2 + e] = p1, pc[(st + 1) & 1][2][hf * 2 + e] = p2;
          }
        // make the next split depend on this one's input cheaply (keeps the compiler from hoisting it out of the loop)
        v[0] = v[0] * 1.0001f;
        v[1] = v[1] + v[0];
      }
      if (INTERLEAVE && WHAT == 3) {
#pragma unroll
        for (int q = 0; q < 6 * NCH; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, DOT ? (NCH == 2 ? 3 : 6) : (NCH == 2 ? 5 : 9), 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 8; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int sl = 0; sl < 2; ++sl)
    for (int i = 0; i < 3; ++i) s += (float)pc[sl][i][0] + (float)pc[sl][i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + v[0][0] + v[1][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int WHAT, int IL, int NCH, int DOT = 0>
static void run(int threads, float* d, long long* dc) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WHAT, IL, NCH, DOT>), dim3(256), dim3(threads), 0, 0, d, 10, 1.f, dc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WHAT, IL, NCH, DOT>), dim3(256), dim3(threads), 0, 0, d, iters, 1.f, dc);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long c;
  (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  static const char* wn[] = {"", "split only (8 values: 44 VALU)", "6*chains MFMA only", "6*chains MFMA + split"};
  printf("%s waves/SIMD=%d chains=%d %-32s interleave=%d  %7.1f ns per step per wave (wall)  %7.1f counter ticks per step\n", DOT ? "dot2c" : "trunc", threads / 256, NCH, wn[WHAT], IL,
         ms * 1e6 / (iters * 4.0), (double)c / (iters * 4.0));
}

int main() {
  float* d;
  long long* dc;
  (void)hipMalloc(&d, 512 * 512 * 4);
  (void)hipMalloc(&dc, 8);
  for (int threads : {256, 512}) {
    run<1, 0, 1>(threads, d, dc);
    run<2, 0, 1>(threads, d, dc);
    run<3, 0, 1>(threads, d, dc);
    run<3, 1, 1>(threads, d, dc);
    run<2, 0, 2>(threads, d, dc);
    run<3, 0, 2>(threads, d, dc);
    run<3, 1, 2>(threads, d, dc);
    run<1, 0, 1, 1>(threads, d, dc);
    run<3, 0, 1, 1>(threads, d, dc);
    run<3, 1, 1, 1>(threads, d, dc);
    run<3, 0, 2, 1>(threads, d, dc);
    run<3, 1, 2, 1>(threads, d, dc);
  }
  return 0;
}
