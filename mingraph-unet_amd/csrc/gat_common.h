// Shared by gat.hip and gat_fused.hip: order-preserving float <-> unsigned encoding and the SLOTTED per-(graph, head) max
// accumulators.  torch.max(e) over all edges of a graph (graph_attention.py:86) is reduced with atomicMax; thousands of
// wavefronts hitting the 32 words of one 128-byte line (8 graphs x 4 heads) serialise in L2 at ~7 ns each -- 29 us for the
// 8 192-node headline batch, whatever the kernel around them does.  So the accumulator is [entry][GMAX_SLOTS]: a wave adds to
// slot (its global wave id % 64) of its entry -- the atomics of an entry spread over four lines and those of different entries
// never share one -- and a reader takes the max over the entry's 64 slots with ONE coalesced 512-byte load (lane = slot) and a
// 6-step wave reduction.  (`gstride` is kept in the signatures for the slot-major layout this replaced: a reader then touched
// 64 different lines per load, which cost the gather kernel 9 us at 64 graphs.)
#pragma once
#include <hip/hip_runtime.h>

namespace mgu {

constexpr int GMAX_SLOTS = 64;

__device__ __forceinline__ unsigned gat_enc_ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float gat_dec_ordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
// An accumulator word is (generation << 32) | encoded value: every layer call draws a new generation (host counter), so
// whatever an earlier call left behind is smaller than anything this call adds and nothing ever needs clearing (a scheme
// with two alternating arrays, each cleared by the other's last kernel, lost updates when calls of different sizes
// interleaved).  A word of an older generation reads as "no edge".
typedef unsigned long long gmax_t;
__device__ __forceinline__ void gmax_add(gmax_t* __restrict__ gmax, int gstride, int slot, int e, unsigned gen, float v) {
  atomicMax(&gmax[(size_t)e * GMAX_SLOTS + slot], ((gmax_t)gen << 32) | gat_enc_ordered(v));
}
__device__ __forceinline__ float gmax_decode(gmax_t w, unsigned gen) {
  return (unsigned)(w >> 32) == gen ? gat_dec_ordered((unsigned)w) : -INFINITY;
}
// Wave-wide max of an unsigned: four DPP steps fold each row of 16 lanes (quad permutes, row_half_mirror, row_mirror), four
// v_readlane + scalar max join the rows.  (A __shfl_xor butterfly is six dependent ds_bpermute round trips through the LDS
// crossbar: 2 us per wave for the four heads of the gather kernel, 9 us on the whole kernel at 64 graphs.)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float wave_max_f32(float v) {   // -inf for "nothing"
  return gat_dec_ordered(wave_max_u32(v > -INFINITY ? gat_enc_ordered(v) : 0u));
}
// max over the slots of entry `e` (wave-uniform), computed by the whole wave: every lane returns the same value
__device__ __forceinline__ float gmax_read_wave(const gmax_t* __restrict__ gmax, int gstride, int e, unsigned gen) {
  const gmax_t w = gmax[(size_t)e * GMAX_SLOTS + (threadIdx.x & 63)];   // the entry's 64 slots are 512 contiguous bytes
  const unsigned u = wave_max_u32((unsigned)(w >> 32) == gen ? (unsigned)w : 0u);   // stale generations count as nothing
  return u ? gat_dec_ordered(u) : -INFINITY;
}
// the same for a per-lane entry (tiles that straddle two graphs: rare)
__device__ __forceinline__ float gmax_read_lane(const gmax_t* __restrict__ gmax, int gstride, int e, unsigned gen) {
  gmax_t u = 0;
  for (int sl = 0; sl < GMAX_SLOTS; ++sl) {
    const gmax_t o = gmax[(size_t)e * GMAX_SLOTS + sl];
    u = o > u ? o : u;
  }
  return gmax_decode(u, gen);
}

}  // namespace mgu
