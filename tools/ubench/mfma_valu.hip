// Micro-benchmark: what hides under an MFMA on gfx950?  Each wave loops over 8 MFMAs (4 independent accumulators) with V
// other instructions placed after every MFMA.  Prints the time per MFMA per SIMD; the pure-MFMA rows give the issue rate
// (v_mfma_f32_32x32x2_f32: 64 cycles; v_mfma_f32_32x32x16_bf16: 32 cycles).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { FMA = 0, PKFMA = 1, IADD = 2, DSREAD = 3, PKADD = 4, NONE = 5 };

template <int MF, int KIND, int V>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  __shared__ f32x4 lds[1024];
  lds[threadIdx.x] = f32x4{seed, seed, seed, seed};
  lds[threadIdx.x + 512] = f32x4{seed, seed, seed, seed};
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 2.f;
  bf16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(seed + i); bh[i] = (__bf16)(seed - i); }
  float x[16];
  f32x2 p[16];
  int n[16];
  f32x4 dv[8];
  for (int i = 0; i < 16; ++i) { x[i] = seed + i + threadIdx.x; p[i] = f32x2{x[i], x[i]}; n[i] = i + threadIdx.x; }
  for (int i = 0; i < 8; ++i) dv[i] = f32x4{0, 0, 0, 0};
  const float c1 = 1.0001f, c2 = 0.5f;
  const f32x2 q1 = {1.0001f, 1.0001f}, q2 = {0.5f, 0.5f};
  const unsigned la = (threadIdx.x & 511) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(ah), "v"(bh));
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int i = (m * V + v) & 15;
        if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
        if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q1), "v"(q2));
        if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q2));
        if (KIND == IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));
        if (KIND == DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(dv[i & 7]) : "v"(la));
      }
    }
    if (KIND == DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 16; ++i) s += x[i] + p[i][0] + p[i][1] + n[i];
  for (int i = 0; i < 8; ++i) s += dv[i][0] + dv[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_base[2][2];
template <int MF, int KIND, int V>
static void run(int threads, float* d) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, KIND, V>), dim3(256), dim3(threads), 0, 0, d, 10, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MF, KIND, V>), dim3(256), dim3(threads), 0, 0, d, iters, 1.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;
  const double ns = ms * 1e6 / (wps * iters * 8.0);
  static const char* kn[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "ds_read_b128", "v_pk_add_f32", "none"};
  const int ti = threads == 512;
  if (V == 0) g_base[MF][ti] = ns;
  printf("%-6s waves/SIMD=%d  %-13s V=%2d  %6.2f ns per MFMA per SIMD", MF ? "bf16" : "f32", threads / 256, kn[KIND], V, ns);
  if (V) printf("   +%.2f ns per extra instr (per wave)", (ns - g_base[MF][ti]) / V / 1.0);
  printf("\n");
}

template <int MF>
static void sweep(int threads, float* d) {
  run<MF, NONE, 0>(threads, d);
  run<MF, FMA, 2>(threads, d);
  run<MF, FMA, 4>(threads, d);
  run<MF, FMA, 8>(threads, d);
  run<MF, PKFMA, 2>(threads, d);
  run<MF, PKFMA, 4>(threads, d);
  run<MF, PKFMA, 8>(threads, d);
  run<MF, PKADD, 4>(threads, d);
  run<MF, IADD, 4>(threads, d);
  run<MF, IADD, 8>(threads, d);
  run<MF, DSREAD, 1>(threads, d);
  run<MF, DSREAD, 2>(threads, d);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 512 * 512 * 4);
  for (int threads : {256, 512}) {
    sweep<0>(threads, d);
    sweep<1>(threads, d);
  }
  return 0;
}
