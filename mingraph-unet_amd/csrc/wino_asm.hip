// Host side of the hand-scheduled assembly form of the wide component-pair Winograd kernel (asm/gen_wino_cp.py emits the
// kernel, the Makefile assembles it into a gfx950 code object and embeds it in the library as a byte array).
//
// Same operator, same layouts and the same arithmetic order as wino3x3_cp_kernel<2, false, false, false> (wino_f32.hip; the
// 3x3 convolutions of ConvBlock, model/unet/unet_encoder.py:15-25): outputs are bitwise equal to the C++ kernel's
// (tests/test_gpu_wino_asm.py).  The code object is loaded once per device through the module API; everything the assembly
// does not cover (ragged sizes, statistics epilogue, odd chunk counts, missing scale / shift) stays on the C++ kernel.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <mutex>

extern "C" const unsigned char mgu_wino_cp2_hsaco[];
extern "C" const unsigned mgu_wino_cp2_hsaco_len;

namespace mgu {

namespace {
struct WinoAsmArgs {          // kernarg segment of mgu_wino_cp2_gfx950 (asm/gen_wino_cp.py: emit_prologue)
  const float* in;            //   0
  const void* wu;             //   8
  float* out;                 //  16  d.out + d.coff
  const float* scale;         //  24
  const float* shift;         //  32
  float* pool;                //  40  may be null
  int H, W, ldin, ldout;      //  48
  int ldpool, nC, relu, tiles_x;      //  64
  int tiles_y, total, ppb, ngroups;   //  80
  int nitems, per_xcd;                //  96
  unsigned mg_ngroups, mg_tx;         // 104  floor(2^32 / d) (0xffffffff for d == 1)
  unsigned mg_txty, pad;              // 112
};
static_assert(sizeof(WinoAsmArgs) == 120, "kernarg layout of mgu_wino_cp2_gfx950");

unsigned magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / d); }

struct Plan {
  int tiles_x, tiles_y, total, nblk, ppb, ngroups, per_xcd;
};
bool is_wide(const IgemmDesc& d) { return d.N > 32 && tun(d).wino_mode != 1; }
Plan plan_of(const IgemmDesc& d) {
  Plan p;
  p.tiles_x = (d.W + 31) / 32, p.tiles_y = (d.H + 7) / 8;
  const int B = d.M / (d.H * d.W);
  const int nc = is_wide(d) ? 64 : 32;   // output channels of a workgroup
  p.total = p.tiles_x * p.tiles_y * B, p.nblk = (d.N + nc - 1) / nc;
  const int rounds = std::max(1, tun(d).wino_rounds), cap = std::max(1, tun(d).wino_ppb_cap);
  int ppb = (int)(((long)p.total * p.nblk) / (256 * rounds));   // the C++ launcher's walk (launch_wino_cp)
  if (ppb < 1) ppb = 1;
  if (ppb > cap) ppb = cap;
  p.ppb = ppb;
  p.ngroups = (p.total + ppb - 1) / ppb;
  p.per_xcd = (p.ngroups * p.nblk + 7) / 8;
  return p;
}

constexpr int MAX_VARIANTS = 32;   // wino_asm = 1: the shipping kernel; n > 1: timing-only variant _v(n-1) of a GEN_WINO_VARIANTS=1 build
std::mutex g_mu;
hipModule_t g_mod[64] = {};
hipFunction_t g_fn[64][MAX_VARIANTS + 2] = {};   // loaded functions per device (the last two: the narrow kernels), written once under
                                                 // g_mu, immutable afterwards
}  // namespace

bool wino_asm_applicable(const IgemmDesc& d) {
  if (!tun(d).wino_asm || !tun(d).wino_prec || !tun(d).wino_cp || tun(d).wino_yfast || tun(d).wino_prio) return false;
  if (d.stat_slots) return false;   // (a missing scale / shift array is 1 / 0 in the kernels, as in the C++ epilogue)
  if (is_wide(d)) {
    if (d.N & 63) return false;
  } else {   // narrow kernels: exactly one 32-channel tile, 2 or 4 chunks (their weight pieces stay in registers), the C++ kernel's
             // two-chunk load lead and reader-side scale / shift (what launch_wino_f32 picks for these layers)
    if (!tun(d).wino_asm_narrow || d.N != 32 || !(d.Cp == 32 || d.Cp == 64) || !tun(d).wino_cp_narrow || !tun(d).wino_deep || tun(d).wino_asm > 1) return false;
  }
  if ((d.H & 7) || (d.W & 31) || (d.Cp & 31) || (d.ldin & 3) || (d.ldout & 3) || (d.coff & 3)) return false;
  if (d.pool && ((d.ldpool & 3) || (d.H & 1) || (d.W & 1))) return false;
  if ((long)d.H * d.W * d.ldin * 4 >= 0x7fff0000l || (long)d.H * d.W * d.ldout * 4 >= 0x7fff0000l) return false;
  const Plan p = plan_of(d);
  if ((long)p.total * p.tiles_x * p.tiles_y >= (1l << 32) || (long)p.ngroups * p.nblk * p.ngroups >= (1l << 32)) return false;
  return true;
}

hipError_t launch_wino_cp_asm(const IgemmDesc& d, hipStream_t s) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  hipFunction_t fn;
  const bool wide = is_wide(d);
  const int var = wide ? std::min(std::max(tun(d).wino_asm, 1), MAX_VARIANTS) - 1 : MAX_VARIANTS + (d.Cp == 64);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_mod[dev]) {
      e = hipModuleLoadData(&g_mod[dev], mgu_wino_cp2_hsaco);
      if (e != hipSuccess) return e;
    }
    if (!g_fn[dev][var]) {
      char name[64];
      if (!wide) snprintf(name, sizeof name, "mgu_wino_cp1r%d_gfx950", d.Cp >> 4);
      else if (var) snprintf(name, sizeof name, "mgu_wino_cp2_gfx950_v%d", var);
      else snprintf(name, sizeof name, "mgu_wino_cp2_gfx950");
      e = hipModuleGetFunction(&g_fn[dev][var], g_mod[dev], name);
      if (e != hipSuccess) return e;
    }
    fn = g_fn[dev][var];
  }
  const Plan p = plan_of(d);
  WinoAsmArgs a;
  a.in = d.in, a.wu = d.wu, a.out = d.out + d.coff, a.scale = d.scale, a.shift = d.shift, a.pool = d.pool;
  a.H = d.H, a.W = d.W, a.ldin = d.ldin, a.ldout = d.ldout;
  a.ldpool = d.pool ? d.ldpool : 0, a.nC = d.Cp >> 4, a.relu = d.relu, a.tiles_x = p.tiles_x;
  a.tiles_y = p.tiles_y, a.total = p.total, a.ppb = p.ppb, a.ngroups = p.ngroups;
  a.nitems = p.ngroups * p.nblk, a.per_xcd = p.per_xcd;
  a.mg_ngroups = magic((unsigned)p.ngroups), a.mg_tx = magic((unsigned)p.tiles_x);
  a.mg_txty = magic((unsigned)(p.tiles_x * p.tiles_y)), a.pad = 0;
  size_t sz = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)(8 * p.per_xcd), 1, 1, 512, 1, 1, 0, s, nullptr, extra);
}

}  // namespace mgu
