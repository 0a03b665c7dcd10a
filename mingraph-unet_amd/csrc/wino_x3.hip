// Winograd F(2x2, 3x3) convolution with fp32 operands split into three exact bf16 pieces, on the bf16 matrix cores of
// gfx950 (CDNA4).  Same operator and same numerics class as wino_f32.hip (fp32 in, fp32 accumulate, fp32 out; the
// product a b is formed from the six piece products of weight >= 2^-16, the dropped ones are below one fp32 rounding),
// same U layout (pack_wino_w_kernel, prec = 1), same epilogue (folded scale/shift/ReLU, fused MaxPool2d(2), fused
// BatchNorm batch statistics).  What changes is the machine mapping, because the cost structure is different:
//
//   * six v_mfma_f32_32x32x16_bf16 (6 x 32 cycles) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles) per 16 input
//     channels, and -- measured, tools/ubench/mfma_valu.hip -- the fp32 MFMA blocks the VALU while it runs whereas the
//     bf16 MFMA lets up to ~4 VALU instructions per MFMA issue for free.  The input transform + split (7.5 VALU per V
//     value) therefore has to be done ONCE per V value and hidden under the MFMAs that consume it;
//   * so a workgroup is FOUR wavefronts, one per SIMD with the whole 512-register file: wave i owns row i of the 4x4
//     transform for all 64 tiles of the 8 x 32 pixel patch and all 32*NT output channels (NT = 2: 256 accumulator
//     registers).  No V value and no U fragment is formed or loaded twice in the workgroup;
//   * a 16-channel chunk is eight steps (component j, m tile): the 6*NT MFMAs of step s are interleaved
//     (sched_group_barrier) with the split of step s+1, with a quarter of the NEXT chunk's LDS reads + transform, and
//     with the prefetch of the next component's U pieces -- nothing of a chunk is waited for in the chunk it is issued;
//   * the raw halo is double buffered in LDS with one barrier per chunk: in chunk c the waves read chunk c+1 (transform
//     ahead) and park chunk c+2, so the buffer written is the one whose reads ended a barrier ago.
//
// STATUS (round 1, measured on MI355X): EXPERIMENTAL, opt-in with MGU_WINO_PREC=2, numerically verified (the GPU kernel
// and parity suites pass with it) but SLOWER than the eight-wavefront kernels of wino_f32.hip: 5.3 ms per headline step
// against 4.05 ms (fp32 MFMA) and 3.77 ms (MGU_WINO_PREC=1).  Two measured reasons:
//   * a CU issues at most one instruction per wavefront every 4 cycles, so with ONE wave per SIMD the ~800 instructions
//     of a chunk (536 VALU, 96 MFMA, LDS, loads, scalar) take >= 3300 cycles by issue alone, and dependent VALU chains
//     (and -> sub -> and -> sub -> perm) stretch that to ~6000: removing every MFMA from the loop does not make it
//     faster.  The VALU work needs a second wave per SIMD to fill the issue slots;
//   * 256 accumulator registers + 3 x bf16 operands leave no room: the NT = 2 epilogue spills (40 us per patch).
// The next design step is recorded in DESIGN.md (waves split by component pair j, two per SIMD).
//
// Reference: model/unet/unet_encoder.py:15-25, unet_decoder.py:22-33 (ConvBlock convs), through IgemmDesc like every
// other conv kernel here.
#include "common.h"
#include <type_traits>

namespace mgu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exact three-way split of two fp32 values into packed bf16 pieces (low half: a, high half: b)
__device__ __forceinline__ void split3_pack(const float a, const float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// An opaque use + redefinition of a value: IR-level code motion (sinking a computation into the block of its first use,
// i.e. behind the next barrier) cannot move the computation past this point.  No instruction is emitted.
template <class T>
__device__ __forceinline__ void pin(T& x) { asm volatile("" : "+v"(x)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

}  // namespace

template <int NT>
__global__ __launch_bounds__(256) void wino3x3_x3_kernel(const IgemmDesc d, const int tiles_x, const int tiles_y, const int total_patches,
                                                         const int patches_per_block, const int ngroups, const int nitems,
                                                         const int per_xcd) {
  constexpr int NWAVES = 4;
  constexpr int NC = 32 * NT;                    // output channels per workgroup
  constexpr int ZP = NC + 8;                     // exchange-buffer pitch of a tile (floats)
  constexpr int QPT = NC / 4;                    // channel quads per tile
  constexpr int UPT = 64 * QPT / 256;            // (tile, channel quad) units a thread finishes per pass
  constexpr int RW = 34, HPIX = 10 * RW;         // raw halo of the 8 x 32 pixel patch
  constexpr int PLD = 20;                        // floats per raw pixel in LDS (16 channels + 4 pad), as in wino_f32.hip
  constexpr int HSTRIDE = NWAVES * 16;           // raw pixels staged per pass (4 threads x 16 bytes per pixel)
  constexpr int HR = (HPIX + HSTRIDE - 1) / HSTRIDE;
  constexpr int S1 = 17 * PLD, S2 = PLD, S3 = 17 * PLD + PLD;   // LDS offsets of tile columns 1..3 (parity planes)
  constexpr int RAWF = HR * HSTRIDE / RW * 34 * PLD + 34 * PLD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zx = smem + 2 * RAWF;      // [4 rows i][64 tiles][ZP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);   // transform row of this wave
  const int lr = lane & 31, lh = lane >> 5;
  const int tx = lr & 15, ty = lr >> 4;
  const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);   // XCD-aware order, as wino_f32.hip
  if (item >= nitems) return;
  const int nblock = item / ngroups;
  const int p_begin = (item - nblock * ngroups) * patches_per_block;
  const int npatch = min(patches_per_block, total_patches - p_begin);
  if (npatch <= 0) return;

  // row i of B^T d:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const float sgn = wi == 1 ? 1.f : -1.f;
  int offA[2], offB[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int rowbase = 2 * (2 * mt + ty);
    offA[mt] = ((rowbase + ra) * 34 + tx) * PLD + lh * 8;
    offB[mt] = ((rowbase + rb) * 34 + tx) * PLD + lh * 8;
  }
  const int nC = d.Cp >> 4;                        // 16-channel chunks
  // U pieces: [ntile][chunk][i*4+j][piece][lane], 16 bytes per lane
  const u32x4* const upx = reinterpret_cast<const u32x4*>(d.wu) + ((size_t)(nblock * NT) * nC * 16 + wi * 4) * 192 + lane;
  const size_t nt_stride = (size_t)nC * 16 * 192;

  // ---- raw halo staging: thread -> (pixel hp0 + HSTRIDE i, 16-byte piece kq) ----
  const int kq = tid & 3, hp0 = tid >> 2;
  int hoff[HR];
  unsigned hmask = 0u, hmask_next = 0u;
  const float* load_base = d.in;
  auto setup_patch = [&](int p, int& img, int& y0, int& x0) {
    const int px = p % tiles_x;
    const int py = (p / tiles_x) % tiles_y;
    img = p / (tiles_x * tiles_y);
    y0 = py * 8;
    x0 = px * 32;
  };
  auto setup_load = [&](int p) {
    int img, y0, x0;
    setup_patch(p, img, y0, x0);
    load_base = d.in + (size_t)img * d.H * d.W * d.ldin + kq * 4;
    unsigned mk = 0u;
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HSTRIDE * i;
      const int r = hp / RW, cc = hp - r * RW;
      const int y = y0 - 1 + r, x = x0 - 1 + cc;
      const bool ok = hp < HPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
      hoff[i] = ok ? (y * d.W + x) * d.ldin : 0;   // unconditional loads from a mapped address; zeroed at the LDS store
      mk |= ok ? (1u << i) : 0u;
    }
    hmask_next = mk;
  };
  f32x4 hreg[HR];
  auto load_halo = [&](int c) {
    hmask = hmask_next;
#pragma unroll
    for (int i = 0; i < HR; ++i) hreg[i] = *reinterpret_cast<const f32x4*>(load_base + hoff[i] + c * 16);
  };
  auto store_halo = [&](float* Hs) {
#pragma unroll
    for (int i = 0; i < HR; ++i) {
      const int hp = hp0 + HSTRIDE * i;
      const int r = hp / RW, cc = hp - r * RW;
      *reinterpret_cast<f32x4*>(Hs + ((r * 2 + (cc & 1)) * 17 + (cc >> 1)) * PLD + kq * 4) =
          ((hmask >> i) & 1u) ? hreg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // the load stream runs three chunks ahead of the MFMAs: (lp, lc) = patch / chunk of the NEXT load.  Loads are
  // unconditional (past the end of the stream the last patch is re-read and never used).
  int lp = 0, lc = 0;
  auto prep_next = [&]() {
    if (lc == 0 && lp < npatch) setup_load(p_begin + lp);
  };
  auto load_next = [&]() {
    load_halo(lc);
    lc = lc + 1 == nC ? 0 : lc + 1;
    lp += lc == 0 ? 1 : 0;
  };

  // finishing role of this thread in the epilogue: channel quad cq of the workgroup's NC channels
  const int cq = tid % QPT;
  const int n0 = nblock * NC + cq * 4;
  const bool fast_n = (d.N % NC == 0) && (d.ldout % 4 == 0) && (d.coff % 4 == 0) && (!d.pool || d.ldpool % 4 == 0);
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n0 + e < d.N) {
      if (d.scale) sc4[e] = d.scale[n0 + e];
      if (d.shift) sh4[e] = d.shift[n0 + e];
    }

  u32x4 bx[2][NT][3];   // U pieces of component j (slot j & 1) and of the next one
  auto load_bx = [&](const int slot, const int j, const int chunk) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) bx[slot][nt][pc] = upx[nt * nt_stride + (size_t)chunk * (16 * 192) + j * 192 + pc * 64];
  };

  f32x16 acc[4][2][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][mt][nt][r] = 0.f;

  f32x4 v[2][2][4][2];   // [chunk parity][m tile][j][channel half]: V[i][j] of this lane's tile, channels 8h .. 8h+7
  unsigned pc[2][3][4];  // [step parity][piece][channel pair]: packed A operand of a step
  auto transform = [&](const float* Hs, auto par_c, auto mt_c, auto hf_c) {
    constexpr int par = decltype(par_c)::value, mt = decltype(mt_c)::value, hf = decltype(hf_c)::value;
    const float* pa = Hs + offA[mt] + hf * 4;
    const float* pb = Hs + offB[mt] + hf * 4;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(pa) + sgn * *reinterpret_cast<const f32x4*>(pb);
    const f32x4 r1 = *reinterpret_cast<const f32x4*>(pa + S1) + sgn * *reinterpret_cast<const f32x4*>(pb + S1);
    const f32x4 r2 = *reinterpret_cast<const f32x4*>(pa + S2) + sgn * *reinterpret_cast<const f32x4*>(pb + S2);
    const f32x4 r3 = *reinterpret_cast<const f32x4*>(pa + S3) + sgn * *reinterpret_cast<const f32x4*>(pb + S3);
    v[par][mt][0][hf] = r0 - r2;
    v[par][mt][1][hf] = r1 + r2;
    v[par][mt][2][hf] = r2 - r1;
    v[par][mt][3][hf] = r1 - r3;
  };
  auto split = [&](auto par_c, auto mt_c, auto j_c, auto slot_c) {
    constexpr int par = decltype(par_c)::value, mt = decltype(mt_c)::value, j = decltype(j_c)::value, slot = decltype(slot_c)::value;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        unsigned p0, p1, p2;
        split3_pack(v[par][mt][j][hf][2 * e], v[par][mt][j][hf][2 * e + 1], p0, p1, p2);
        pc[slot][0][hf * 2 + e] = p0, pc[slot][1][hf * 2 + e] = p1, pc[slot][2][hf * 2 + e] = p2;
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue: chunk 0 in LDS and transformed, chunk 1 parked, chunk 2 in flight ----
  prep_next();
  load_next();
  load_bx(0, 0, 0);
  store_halo(smem);
  prep_next();
  load_next();
  lds_barrier();
  transform(smem, I0{}, I0{}, I0{});
  transform(smem, I0{}, I0{}, I1{});
  transform(smem, I0{}, I1{}, I0{});
  transform(smem, I0{}, I1{}, I1{});
  split(I0{}, I0{}, I0{}, I0{});
  store_halo(smem + RAWF);
  prep_next();
  load_next();

  f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};   // training: sum z, sum z^2 of this thread's channel quad
  int rbuf = 1;   // raw buffer holding the chunk AFTER the one being multiplied

  // one 16-channel chunk: eight steps (j = s >> 1, m tile = s & 1)
  // ---- the pieces of work that ride behind the MFMAs, each a handful of instructions -------------------------------
  // transform unit (m tile, half) of the next chunk: raw reads, row part, column part
  f32x4 rwa[4], rwb[4], rr[4];
  auto t_read = [&](const float* Hs, auto mt_c, auto hf_c, auto x_c) __attribute__((always_inline)) {
    constexpr int mt = decltype(mt_c)::value, hf = decltype(hf_c)::value, x = decltype(x_c)::value;
    constexpr int SX[4] = {0, S1, S2, S3};
    rwa[x] = *reinterpret_cast<const f32x4*>(Hs + offA[mt] + hf * 4 + SX[x]);
    rwb[x] = *reinterpret_cast<const f32x4*>(Hs + offB[mt] + hf * 4 + SX[x]);
  };
  auto t_row = [&](auto x_c) __attribute__((always_inline)) {
    constexpr int x = decltype(x_c)::value;
    rr[x] = rwa[x] + sgn * rwb[x];
    pin(rr[x]);
  };
  auto t_col = [&](auto par_c, auto mt_c, auto hf_c, auto j_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value, mt = decltype(mt_c)::value, hf = decltype(hf_c)::value, j = decltype(j_c)::value;
    v[par][mt][j][hf] = j == 0 ? rr[0] - rr[2] : (j == 1 ? rr[1] + rr[2] : (j == 2 ? rr[2] - rr[1] : rr[1] - rr[3]));
    pin(v[par][mt][j][hf]);
  };
  // split of one value pair (p = half * 2 + e) of V[.][j] into its packed pieces, in two halves
  float sra[4], srb[4];
  auto split_a = [&](auto par_c, auto mt_c, auto j_c, auto slot_c, auto p_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value, mt = decltype(mt_c)::value, j = decltype(j_c)::value, slot = decltype(slot_c)::value,
                  p = decltype(p_c)::value;
    const float a = v[par][mt][j][p >> 1][2 * (p & 1)], b = v[par][mt][j][p >> 1][2 * (p & 1) + 1];
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    pc[slot][0][p] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    pc[slot][1][p] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    sra[p] = ra, srb[p] = rb;
    pin(pc[slot][0][p]), pin(pc[slot][1][p]), pin(sra[p]), pin(srb[p]);
  };
  auto split_b = [&](auto slot_c, auto p_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value, p = decltype(p_c)::value;
    const float ra = sra[p], rb = srb[p];
    const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
    pc[slot][2][p] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    pin(pc[slot][2][p]);
  };
  auto store_one = [&](float* Hs, auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    const int hp = hp0 + HSTRIDE * i;
    const int r = hp / RW, cc = hp - r * RW;
    *reinterpret_cast<f32x4*>(Hs + ((r * 2 + (cc & 1)) * 17 + (cc >> 1)) * PLD + kq * 4) =
        ((hmask >> i) & 1u) ? hreg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto load_one = [&](auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    hreg[i] = *reinterpret_cast<const f32x4*>(load_base + hoff[i] + lc * 16);
  };
  auto load_done = [&]() __attribute__((always_inline)) {
    lc = lc + 1 == nC ? 0 : lc + 1;
    lp += lc == 0 ? 1 : 0;
  };
  auto load_b1 = [&](const int slot, const int j, const int chunk, auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value, nt = i / 3, pcx = i % 3;
    bx[slot][nt][pcx] = upx[nt * nt_stride + (size_t)chunk * (16 * 192) + j * 192 + pcx * 64];
  };

  // one 16-channel chunk: eight steps (j = s >> 1, m tile = s & 1) of 6 NT MFMAs.  The source order below IS the
  // schedule: every MFMA is followed by its share of the other work and a scheduling fence.
  auto chunk = [&](auto par_c, const int cc) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;
    const float* Hn = smem + rbuf * RAWF;           // chunk c+1: transformed ahead
    float* Hw = smem + (rbuf ^ 1) * RAWF;           // chunk c+2 is parked here
    const int cn = cc + 1 == nC ? 0 : cc + 1;
    static_for<0, 8>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int j = s >> 1, mt = s & 1, slot = s & 1;
      // operands being prepared: the next step's (step 0 of the next chunk after the last step)
      using NPar = std::integral_constant<int, (s < 7 ? P : (P ^ 1))>;
      using NMt = std::integral_constant<int, (s < 7 ? ((s + 1) & 1) : 0)>;
      using NJ = std::integral_constant<int, (s < 7 ? ((s + 1) >> 1) : 0)>;
      using NSlot = std::integral_constant<int, ((s + 1) & 1)>;
      // transform unit of the next chunk handled in this step (steps 4 .. 7): (m tile, half) = (0,0) (0,1) (1,0) (1,1)
      using UPar = std::integral_constant<int, (P ^ 1)>;
      using UMt = std::integral_constant<int, ((s >> 1) & 1)>;
      using UHf = std::integral_constant<int, (s & 1)>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // the six piece products, smallest first
      constexpr int NM = 6 * NT;
      static_for<0, NM>([&](auto k_c) {
        constexpr int k = decltype(k_c)::value;
        constexpr int q = k / NT, nt = k % NT;
        acc[j][mt][nt] = mfma_bf16(u32x4{pc[slot][PA[q]][0], pc[slot][PA[q]][1], pc[slot][PA[q]][2], pc[slot][PA[q]][3]}, bx[j & 1][nt][PB[q]],
                                   acc[j][mt][nt]);
        if constexpr (NT == 2) {
          if constexpr (k < 8) {
            if constexpr ((k & 1) == 0) split_a(NPar{}, NMt{}, NJ{}, NSlot{}, std::integral_constant<int, (k >> 1)>{});
            else split_b(NSlot{}, std::integral_constant<int, (k >> 1)>{});
          }
          if constexpr (s >= 4) {
            if constexpr (k == 4) t_read(Hn, UMt{}, UHf{}, I0{}), t_read(Hn, UMt{}, UHf{}, I1{});
            if constexpr (k == 5) t_read(Hn, UMt{}, UHf{}, I2{}), t_read(Hn, UMt{}, UHf{}, I3{});
            if constexpr (k == 8) t_row(I0{}), t_row(I1{});
            if constexpr (k == 9) t_row(I2{}), t_row(I3{});
            if constexpr (k == 10) t_col(UPar{}, UMt{}, UHf{}, I0{}), t_col(UPar{}, UMt{}, UHf{}, I1{});
            if constexpr (k == 11) t_col(UPar{}, UMt{}, UHf{}, I2{}), t_col(UPar{}, UMt{}, UHf{}, I3{});
          }
          if constexpr (mt == 0 && k >= 2 && k < 8)   // U pieces of the next component, two steps ahead of their use
            load_b1((j + 1) & 1, j < 3 ? j + 1 : 0, j < 3 ? cc : cn, std::integral_constant<int, k - 2>{});
          if constexpr (s == 1) {
            if constexpr (k == 8) store_one(Hw, I0{}), store_one(Hw, I1{}), store_one(Hw, I2{});
            if constexpr (k == 9) store_one(Hw, I3{}), store_one(Hw, std::integral_constant<int, 4>{}), store_one(Hw, std::integral_constant<int, 5>{});
            if constexpr (k == 10) {
              hmask = hmask_next;
              load_one(I0{}), load_one(I1{}), load_one(I2{});
            }
            if constexpr (k == 11) {
              load_one(I3{}), load_one(std::integral_constant<int, 4>{}), load_one(std::integral_constant<int, 5>{});
              load_done();
            }
          }
        } else {
          if constexpr (k < 4) {
            split_a(NPar{}, NMt{}, NJ{}, NSlot{}, k_c);
            split_b(NSlot{}, k_c);
          }
          if constexpr (s >= 4) {
            if constexpr (k == 2) t_read(Hn, UMt{}, UHf{}, I0{}), t_read(Hn, UMt{}, UHf{}, I1{});
            if constexpr (k == 3) t_read(Hn, UMt{}, UHf{}, I2{}), t_read(Hn, UMt{}, UHf{}, I3{});
            if constexpr (k == 4) t_row(I0{}), t_row(I1{}), t_row(I2{}), t_row(I3{});
            if constexpr (k == 5)
              t_col(UPar{}, UMt{}, UHf{}, I0{}), t_col(UPar{}, UMt{}, UHf{}, I1{}), t_col(UPar{}, UMt{}, UHf{}, I2{}),
                  t_col(UPar{}, UMt{}, UHf{}, I3{});
          }
          if constexpr (mt == 0 && k >= 1 && k < 4) load_b1((j + 1) & 1, j < 3 ? j + 1 : 0, j < 3 ? cc : cn, std::integral_constant<int, k - 1>{});
          if constexpr (s == 1) {
            if constexpr (k == 4) static_for<0, HR>([&](auto i_c) { store_one(Hw, i_c); });
            if constexpr (k == 5) {
              hmask = hmask_next;
              static_for<0, HR>([&](auto i_c) { load_one(i_c); });
              load_done();
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- inverse transform + epilogue of a patch (as wino_f32.hip: column part in registers, the four row waves meet
  //      in LDS, finishing threads take one tile and four consecutive channels) ----
  auto epilogue = [&](const int pi) __attribute__((always_inline)) {
    int img, y0, x0;
    setup_patch(p_begin + pi, img, y0, x0);
    float* const img_out = d.out + (size_t)img * d.H * d.W * d.ldout + d.coff;
    const unsigned sW = (unsigned)(d.W * d.ldout);
    const bool interior = (y0 + 8 <= d.H) && (x0 + 32 <= d.W) && fast_n;   // block-uniform
    float* pool_out = nullptr;
    if (d.pool) pool_out = d.pool + (size_t)img * (d.H >> 1) * (d.W >> 1) * d.ldpool;
    f32x4 pmax[UPT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float z = q == 0 ? (acc[0][mt][nt][r] + acc[1][mt][nt][r] + acc[2][mt][nt][r])
                                   : (acc[1][mt][nt][r] - acc[2][mt][nt][r] - acc[3][mt][nt][r]);
            const int T = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            Zx[(wi * 64 + T) * ZP + nt * 32 + lr] = z;
          }
      lds_barrier();
#pragma unroll
      for (int k = 0; k < UPT; ++k) {
        const int u = tid + k * 256;
        const int T = u / QPT;
        const float* zp = Zx + T * ZP + cq * 4;
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(zp);
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(zp + 64 * ZP);
        const f32x4 z2 = *reinterpret_cast<const f32x4*>(zp + 128 * ZP);
        const f32x4 z3 = *reinterpret_cast<const f32x4*>(zp + 192 * ZP);
        f32x4 ya = (z0 + z1 + z2) * sc4 + sh4;
        f32x4 yb = (z1 - z2 - z3) * sc4 + sh4;
        if (d.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ya[e] = fmaxf(ya[e], 0.f), yb[e] = fmaxf(yb[e], 0.f);
        }
        const int oy = y0 + 2 * (T >> 4), ox = x0 + 2 * (T & 15) + q;
        if (d.stat_slots) {
          const float ma = (interior || (ox < d.W && oy < d.H)) ? 1.f : 0.f;
          const float mb = (interior || (ox < d.W && oy + 1 < d.H)) ? 1.f : 0.f;
          st1 += ma * ya + mb * yb;
          st2 += ma * ya * ya + mb * yb * yb;
        }
        const unsigned idx = (unsigned)((oy * d.W + ox) * d.ldout + n0);
        if (interior) {
          *reinterpret_cast<f32x4*>(img_out + idx) = ya;
          *reinterpret_cast<f32x4*>(img_out + idx + sW) = yb;
        } else if (ox < d.W) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < d.N) {
              if (oy < d.H) img_out[idx + e] = ya[e];
              if (oy + 1 < d.H) img_out[idx + sW + e] = yb[e];
            }
        }
        if (d.pool) {
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(ya[e], yb[e]);
          if (q == 0) {
            pmax[k] = m;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], pmax[k][e]);
            const int py = oy >> 1, px = ox >> 1;
            if (oy + 1 < d.H && ox < d.W) {
              float* pp = pool_out + (size_t)(py * (d.W >> 1) + px) * d.ldpool + n0;
              if (fast_n) {
                *reinterpret_cast<f32x4*>(pp) = m;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (n0 + e < d.N) pp[e] = m[e];
              }
            }
          }
        }
      }
      lds_barrier();   // Zx is rewritten by the next pass / patch
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][mt][nt][r] = 0.f;
  };

  // The chunk loop is unrolled by two so that the V parity is a compile-time constant (a run-time parity would keep
  // both halves of v[] live across the loop edge); the launcher guarantees an even number of chunks per patch.
  auto step = [&](auto par_c, const int cc) __attribute__((always_inline)) {
    prep_next();
    lds_barrier();   // chunk c+1 is visible in buffer rbuf; every wave has finished reading buffer rbuf ^ 1
    chunk(par_c, cc);
    rbuf ^= 1;
  };
  for (int pi = 0; pi < npatch; ++pi) {
    for (int cc = 0; cc < nC; cc += 2) {
      step(I0{}, cc);
      step(I1{}, cc + 1);
    }
    epilogue(pi);
  }
  if (d.stat_slots) {
    float* red = Zx;   // [256][8]; the raw buffers may still be written by the (unused) tail of the load stream
    lds_barrier();
    *reinterpret_cast<f32x4*>(red + tid * 8) = st1;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = st2;
    lds_barrier();
    if (tid < 8 * QPT) {   // thread -> (which sum, channel quad, element)
      const int which = tid / (4 * QPT), rem = tid - which * 4 * QPT, qd = rem >> 2, e = rem & 3;
      double sum = 0.0;
      for (int k = qd; k < 256; k += QPT) sum += (double)red[k * 8 + which * 4 + e];
      const int n = nblock * NC + qd * 4 + e;
      if (n < d.N) atomicAdd(d.stat_slots + (size_t)(blockIdx.x & 63) * 2 * d.N + which * d.N + n, sum);
    }
  }
}

template <int NT>
static hipError_t launch_x3(const IgemmDesc& d, hipStream_t s) {
  constexpr int NWAVES = 4;
  const int tiles_x = (d.W + 31) / 32, tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int total = tiles_x * tiles_y * B, nblk = (d.N + 32 * NT - 1) / (32 * NT);
  // one workgroup per CU is resident (LDS): keep ~2 rounds of them in the grid, each walking its patches
  int ppb = (int)(((long)total * nblk) / (256 * 2));
  if (ppb < 1) ppb = 1;
  if (ppb > 16) ppb = 16;
  const int ngroups = (total + ppb - 1) / ppb;
  const int per_xcd = (ngroups * nblk + 7) / 8;
  dim3 grid(8 * per_xcd, 1);
  constexpr int HSTRIDE = NWAVES * 16, HR = (340 + HSTRIDE - 1) / HSTRIDE;
  constexpr int RAWF = HR * HSTRIDE / 34 * 34 * 20 + 34 * 20;   // must match the kernel
  const size_t lds = (size_t)(2 * RAWF + 4 * 64 * (32 * NT + 8)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino3x3_x3_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((wino3x3_x3_kernel<NT>), grid, dim3(64 * NWAVES), lds, s, d, tiles_x, tiles_y, total, ppb, ngroups, ngroups * nblk,
                     per_xcd);
  return hipGetLastError();
}

bool wino_x3_applicable(const IgemmDesc& d) { return (d.Cp % 32) == 0; }   // an even number of 16-channel chunks

hipError_t launch_wino_x3(const IgemmDesc& d, hipStream_t s) {
  if (!wino_x3_applicable(d)) return hipErrorInvalidValue;
  return d.N > 32 ? launch_x3<2>(d, s) : launch_x3<1>(d, s);
}

}  // namespace mgu
