// ubench: the Winograd epilogue's LDS exchange writes -- 64 x ds_write_b32 per wave and pass (lane = channel, one 128-byte row per
// half wave, a VGPR address + immediate) against ds_write_addtid_b32 (address = M0 + immediate + 4 * lane: no address VGPR, half
// the cycles on the LDS store path per MI355X_MICROARCH.md "LDS").  512 threads (8 waves, the kernel's workgroup), one workgroup
// per CU, every wave writes its 16 KB share region REP times; checks the addtid image against the ordinary one.
// Measured (MI355X): 1.889 ms (ds_write_b32) vs 0.655 ms (add-TID) for 2000 passes, images identical.  Inside wino3x3_cp_kernel the
// same change is worth less (the share arithmetic and the barrier skew remain): DESIGN.md section 3.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_addtid.hip -o /tmp/lds_addtid && /tmp/lds_addtid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x)                                                                     \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

template <int OFF>
__device__ __forceinline__ void st_addtid(float v, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0 offset:%2" ::"v"(v), "s"(m0v), "n"(OFF) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// MODE 0: ds_write_b32 (row = r, half wave h -> tile r' = (r&3) + 8 (r>>2) + 4 h, as the MFMA C layout has it)
// MODE 1: ds_write_addtid_b32 into the slot order 2 r + h (the two half waves of a register are adjacent rows)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* __restrict__ out, const float* __restrict__ in, int rep, long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 8 waves x 2 shares x 32 tiles x 32 channels (64 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float v[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) v[r] = in[(blockIdx.x * 512 + tid) * 32 + r];
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < rep; ++it) {
    if (MODE == 0) {
      float* p = smem + wave * 2048 + (lane >> 5) * 4 * 32 + (lane & 31);
      asm volatile("" : "+v"(p));
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int rr = r & 15, sh = r >> 4;
        p[sh * 1024 + ((rr & 3) + 8 * (rr >> 2)) * 32] = v[r] + (float)it;
      }
    } else {
      const unsigned m0v = (unsigned)(wave * 8192);
      sfor<0, 32>([&](auto R) {
        constexpr int r = decltype(R)::value;
        constexpr int rr = r & 15, sh = r >> 4;
        st_addtid<(sh * 1024 + 2 * rr * 32) * 4>(v[r] + (float)it, m0v);
      });
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  // image in LOGICAL order [wave][share][tile 0..31][ch]: undo the slot permutation of mode 1
  for (int e = tid; e < 8 * 2048; e += 512) {
    const int w = e >> 11, sh = (e >> 10) & 1, t = (e >> 5) & 31, ch = e & 31;
    int slot = t;
    if (MODE == 1) slot = 2 * ((t & 3) + 4 * (t >> 3)) + ((t >> 2) & 1);
    out[(size_t)blockIdx.x * 16384 + e] = smem[w * 2048 + sh * 1024 + slot * 32 + ch];
  }
}

int main() {
  const int NB = 256, REP = 2000;
  std::vector<float> hin((size_t)NB * 512 * 32);
  for (size_t i = 0; i < hin.size(); ++i) hin[i] = (float)((i * 2654435761u) % 1000) * 0.25f;
  float *in, *o0, *o1;
  long long* cyc;
  CHK(hipMalloc(&in, hin.size() * 4));
  CHK(hipMalloc(&o0, (size_t)NB * 16384 * 4));
  CHK(hipMalloc(&o1, (size_t)NB * 16384 * 4));
  CHK(hipMalloc(&cyc, NB * 8));
  CHK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  std::vector<long long> hc(NB);
  for (int mode = 0; mode < 2; ++mode) {
    for (int pass = 0; pass < 2; ++pass) {
      CHK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(512), 65536, 0, o0, in, REP, cyc);
      else hipLaunchKernelGGL(k<1>, dim3(NB), dim3(512), 65536, 0, o1, in, REP, cyc);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      CHK(hipMemcpy(hc.data(), cyc, NB * 8, hipMemcpyDeviceToHost));
      if (pass) printf("mode %d (%s): %.3f ms, %.0f s_memtime ticks per pass of 8 waves x 32 writes (x 100 MHz -> scale), block 0\n", mode,
                       mode ? "ds_write_addtid_b32" : "ds_write_b32", ms, (double)hc[0] / REP);
    }
  }
  std::vector<float> h0((size_t)NB * 16384), h1((size_t)NB * 16384);
  CHK(hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
  printf("images %s (%zu of %zu differ)\n", bad ? "DIFFER" : "identical", bad, h0.size());
  return bad != 0;
}
