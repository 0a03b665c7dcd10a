#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 assembly of the wide component-pair Winograd kernel.

Emits `mgu_wino_cp2_gfx950`: the same algorithm, data layout and arithmetic ORDER as wino3x3_cp_kernel<2, false, false, false>
(csrc/wino_f32.hip; the 3x3 convolutions of ConvBlock, model/unet/unet_encoder.py:15-25) -- results are bitwise equal to the
C++ kernel -- with the register map and the instruction order of the chunk loop fixed by hand:

  * 256 VGPRs per wave, no scratch: 128 accumulators | 48 weight pieces | 2 x 12 operand pieces | 32 raw operands |
    12 halo registers | 9 lane constants | 3 temporaries;
  * a step issues the eight LDS reads of the NEXT step's raw operands first, then its twelve MFMAs with the transform and the
    three-way split of those operands spread between them (nothing waits for a read it has just issued); the weight pieces of
    the next chunk are requested right behind the last MFMA that uses the register they land in;
  * every s_waitcnt is counted (vmcnt: halo loads, weight pieces and stores retire in order; lgkmcnt: LDS only, no scalar
    loads inside the loops);
  * out-of-image halo pixels ride on the buffer descriptor's range check (offset 0x7fff0000): no mask, no select.

Applicability (the launcher checks; everything else stays on the C++ kernel): inference epilogue (no statistics), H % 8 == 0,
W % 32 == 0, N % 64 == 0, Cp % 32 == 0, channel pitches and offsets % 4 == 0, scale and shift present, x-fastest patch order.

Usage: gen_wino_cp.py OUT.s
"""
import os
import sys

DEBUG = os.environ.get("GEN_WINO_DEBUG", "")
# Options of the kernel being emitted (main() emits the shipping kernel and, with GEN_WINO_VARIANTS=1, timing-only variants):
#   no_epilogue   timing only: no inverse transform / stores (accumulators cleared)
#   no_valu       timing only: no input transform / split in the chunk loop (operand pieces stay constant)
#   no_mfma       timing only: no MFMAs in the chunk loop
#   no_barrier    timing only: no s_barrier in the chunk loop
#   b1_step       step in front of which barrier B1 sits (2 or 3)
#   valu_from     first MFMA gap of a step that carries transform work
OPT = {}
def opt(k, d=None): return OPT.get(k, d)
def TT(i):
    """cold-code temporary i (0 / 1): raw half 1's last registers; in the ping-pong experiment (whole raw set in flight across chunk tops)
    the unused second operand-piece slot"""
    return (198 + i) if OPT.get("pingpong") else (230 + i)

# ---------------------------------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------------------------------
def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"

CFG = {"ntb": 2, "nc": None}          # kernel being emitted: output-channel tiles per workgroup; static chunk count (narrow kernels)
def NTB(): return CFG["ntb"]
def ACC(jj, nt, mi): return ((jj * 2 + nt) * 2 + mi) * 16 if CFG["ntb"] == 2 else (jj * 2 + mi) * 16
# narrow kernels (NTB = 1): the layer's whole weight-piece slice of the wave stays in registers, two halo register sets
def WN(c, jj, p): return 64 + ((c * 2 + jj) * 3 + p) * 4
def HSET(setn, i): return (232 if setn == 0 else 160) + i * 4
def BX(jj, nt, p): return 128 + ((jj * 2 + nt) * 3 + p) * 4
def PC(slot, p): return 176 + (slot * 3 + p) * 4
def RAW(hf, k): return 200 + (hf * 4 + k) * 4          # k: 0 = a(x), 1 = b(x), 2 = a(y), 3 = b(y)
def HREG(i): return 232 + i * 4
VA, VB = 244, 245
VHST = [246, 247, 248]
VHOFF = [249, 250, 251]
VLANE16 = 252
VTID = 253
VT0, VT1 = 230, 231   # temporaries of the cold code (halo-offset setup, epilogue addressing, debug stores): the last two registers of raw half 1,
                      # free at every point those run (narrow kernels: see emit_epilogue_n, which keeps its own temporaries clear of them)
def VMASK(): return 254 if CFG["ntb"] == 2 else 172   # 0xffff0000 in a VGPR and the wave's transform sign (+-1.0) in a VGPR: with all-VGPR VOP2
def VSGN(): return 255 if CFG["ntb"] == 2 else 173    # forms (v_fmac / v_add / v_sub / v_and) two waves of a SIMD issue the transform + split at
                                                      # 2.4 cycles per instruction instead of 4 (tools/ubench/gen_issue_cost.py: mix2 vs mix)

# SGPRs
S_IN, S_WU, S_OUT, S_SCALE, S_SHIFT, S_POOL = 4, 6, 8, 10, 12, 14
S_H, S_W, S_LDIN, S_LDOUT, S_LDPOOL, S_NC, S_RELU, S_TX, S_TY, S_TOTAL, S_PPB, S_NGROUPS = 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27
S_NITEMS, S_PERXCD, S_MGNG, S_MGTX, S_MGTXTY, S_PAD = 28, 29, 30, 31, 32, 33
S_NBLOCK, S_PBEGIN, S_NPATCH, S_LC, S_LP, S_PI, S_C, S_LC64 = 34, 35, 36, 37, 38, 39, 40, 41
S_N64X4 = 42        # (nblock * 64) * 4: byte offset of the workgroup's first channel
S_SUMOFF = 43       # (W * ldout + ldout) * 4
S_INR = 44          # s[44:47] input descriptor
S_UR = 48           # s[48:51] weight-piece descriptor
S_SGN, S_W1, S_MASK, S_PERM, S_NTSTRIDE, S_TXTY, S_IMGB, S_OOB = 52, 53, 54, 55, 56, 57, 58, 59
S_T = [60, 61, 62, 63, 64, 65, 66, 67]
S_WI, S_JP = 68, 69
S_P, S_IMG, S_REM, S_PY, S_PX, S_Y0, S_X0 = 70, 71, 72, 73, 74, 75, 72   # S_X0 shares with S_REM after use? no: keep separate below
S_X0 = 88
S_M = 76            # s[76:77]
S_WO = [[78, 79], [90, 91]]   # weight soffsets [jj][nt] of the chunk being requested
S_OUTR = 80         # s[80:83]
S_POOLR = 84        # s[84:87]
S_OUTIMGB = 89
S_SCR = 92          # s[92:95] scale descriptor
S_SHR = 96          # s[96:99] shift descriptor
S_LD4, S_SW4 = 100, 101

S1F, S2F, S3F = 17 * 20, 20, 17 * 20 + 20     # LDS offsets (floats) of tile columns 1..3
BUFX = 0x20000                                # byte distance of the two raw buffers (slots 0 and 4 of the LDS map)
RAWB = 8192 * 4                               # bytes of an LDS slot
ZBIAS = 57472

out = []
def E(s=""): out.append("\t" + s if s and not s.endswith(":") else s)
def L(s): out.append(s + ":")
_lbl = [0]
def newlabel(p="L"):
    _lbl[0] += 1
    return f".{p}_{_lbl[0]}"


def divmod_magic(n, d, mg, q, r, t0, t1):
    """q = n / d, r = n % d  (scalar; mg = floor(2^32 / d), n * d < 2^32)"""
    E(f"s_mul_hi_u32 s{q}, s{n}, s{mg}")
    E(f"s_mul_i32 s{t0}, s{q}, s{d}")
    E(f"s_sub_u32 s{r}, s{n}, s{t0}")
    E(f"s_add_u32 s{t0}, s{q}, 1")
    E(f"s_sub_u32 s{t1}, s{r}, s{d}")
    E(f"s_cmp_ge_u32 s{r}, s{d}")
    E(f"s_cselect_b32 s{q}, s{t0}, s{q}")
    E(f"s_cselect_b32 s{r}, s{t1}, s{r}")


def patch_coords(p):
    """s_p -> S_IMG, S_Y0, S_X0 (x fastest inside an image)"""
    divmod_magic(p, S_TXTY, S_MGTXTY, S_IMG, S_REM, S_T[6], S_T[7])
    divmod_magic(S_REM, S_TX, S_MGTX, S_PY, S_PX, S_T[6], S_T[7])
    E(f"s_lshl_b32 s{S_Y0}, s{S_PY}, 3")
    E(f"s_lshl_b32 s{S_X0}, s{S_PX}, 5")


S_VHST = S_PAD      # 1: the halo-offset registers hold the patch-independent offsets of an INTERIOR patch (kernarg pad word: 0 at entry)


def setup_load():
    """setup_load(p_begin + lp): input descriptor of the patch and the three halo offsets of the thread.
    The per-lane part (three times: unit -> (row, column) of the 10 x 34 halo, bounds, byte offset; 57 VALU in a loop whose VALU issue is
    the bottleneck) depends on the patch only through the bounds: for an INTERIOR patch (no halo pixel outside the image) the patch's
    position goes into the descriptor's base and the offsets are the same for every such patch, so they are formed once and kept while
    interior patches follow each other (x runs fastest: 14 of 16 patches of a 512-wide row)."""
    E(f"s_add_u32 s{S_P}, s{S_PBEGIN}, s{S_LP}")
    patch_coords(S_P)
    E(f"s_mul_i32 s{S_T[6]}, s{S_IMG}, s{S_IMGB}")
    E(f"s_mul_hi_u32 s{S_T[7]}, s{S_IMG}, s{S_IMGB}")
    E(f"s_add_u32 s{S_INR}, s{S_IN}, s{S_T[6]}")
    E(f"s_addc_u32 s{S_INR + 1}, s{S_IN + 1}, s{S_T[7]}")
    ledge, lvalu, ldone = newlabel("edge"), newlabel("hofs"), newlabel("hdone")
    # edge patch?  (y0, x0 are multiples of 8 / 32, H, W too)
    E(f"s_add_u32 s{S_M}, s{S_Y0}, 8")
    E(f"s_cmp_eq_u32 s{S_M}, s{S_H}")
    E(f"s_cselect_b32 s{S_M + 1}, 1, 0")
    E(f"s_cmp_eq_u32 s{S_Y0}, 0")
    E(f"s_cselect_b32 s{S_M + 1}, 1, s{S_M + 1}")
    E(f"s_add_u32 s{S_M}, s{S_X0}, 32")
    E(f"s_cmp_eq_u32 s{S_M}, s{S_W}")
    E(f"s_cselect_b32 s{S_M + 1}, 1, s{S_M + 1}")
    E(f"s_cmp_eq_u32 s{S_X0}, 0")
    E(f"s_cselect_b32 s{S_M + 1}, 1, s{S_M + 1}")
    E(f"s_sub_u32 s{S_T[6]}, s{S_Y0}, 1")     # y0 - 1
    E(f"s_sub_u32 s{S_T[7]}, s{S_X0}, 1")     # x0 - 1
    E(f"s_cmp_lg_u32 s{S_M + 1}, 0")
    E(f"s_cbranch_scc1 {ledge}")
    # interior: base += ((y0 - 1) * W + (x0 - 1)) * ldin * 4  (< 2^31: the launcher's size check)
    E(f"s_mul_i32 s{S_M}, s{S_T[6]}, s{S_W}")
    E(f"s_add_u32 s{S_M}, s{S_M}, s{S_T[7]}")
    E(f"s_mul_i32 s{S_M}, s{S_M}, s{S_LDIN}")
    E(f"s_lshl_b32 s{S_M}, s{S_M}, 2")
    E(f"s_add_u32 s{S_INR}, s{S_INR}, s{S_M}")
    E(f"s_addc_u32 s{S_INR + 1}, s{S_INR + 1}, 0")
    E(f"s_and_b32 s{S_INR + 1}, s{S_INR + 1}, 0xffff")
    E(f"s_cmp_eq_u32 s{S_VHST}, 1")
    E(f"s_cbranch_scc1 {ldone}")             # the registers already hold the interior offsets
    E(f"s_mov_b32 s{S_T[6]}, 0")
    E(f"s_mov_b32 s{S_T[7]}, 0")
    E(f"s_mov_b32 s{S_VHST}, 1")
    E(f"s_branch {lvalu}")
    L(ledge)
    E(f"s_and_b32 s{S_INR + 1}, s{S_INR + 1}, 0xffff")
    E(f"s_mov_b32 s{S_VHST}, 0")
    L(lvalu)
    for i in range(3):
        d = VHOFF[i]
        E(f"v_lshrrev_b32_e32 v{TT(0)}, 2, v{VTID}")
        if i:
            E(f"v_add_u32_e32 v{TT(0)}, {128 * i}, v{TT(0)}")                    # hp
        E(f"v_mul_u32_u24_e32 v{TT(1)}, 0x788, v{TT(0)}")
        E(f"v_lshrrev_b32_e32 v{TT(1)}, 16, v{TT(1)}")                            # r = hp / 34
        E(f"v_mul_u32_u24_e32 v{d}, 34, v{TT(1)}")
        E(f"v_sub_u32_e32 v{d}, v{TT(0)}, v{d}")                                # cc
        E(f"v_cmp_gt_u32_e32 vcc, 0x154, v{TT(0)}")                             # hp < 340
        E(f"v_add_u32_e32 v{TT(1)}, s{S_T[6]}, v{TT(1)}")                         # y
        E(f"v_add_u32_e32 v{d}, s{S_T[7]}, v{d}")                             # x
        E(f"v_cmp_gt_u32_e64 s[{S_M}:{S_M + 1}], s{S_H}, v{TT(1)}")
        E(f"s_and_b64 vcc, vcc, s[{S_M}:{S_M + 1}]")
        E(f"v_cmp_gt_u32_e64 s[{S_M}:{S_M + 1}], s{S_W}, v{d}")
        E(f"s_and_b64 vcc, vcc, s[{S_M}:{S_M + 1}]")
        E(f"v_mad_u32_u24 v{TT(1)}, v{TT(1)}, s{S_W}, v{d}")                      # y * W + x
        E(f"v_mul_lo_u32 v{TT(1)}, v{TT(1)}, s{S_LDIN}")
        E(f"v_and_b32_e32 v{TT(0)}, 3, v{VTID}")                                # kq
        E(f"v_lshl_add_u32 v{TT(1)}, v{TT(0)}, 2, v{TT(1)}")
        E(f"v_lshlrev_b32_e32 v{TT(1)}, 2, v{TT(1)}")                             # bytes
        E(f"v_mov_b32_e32 v{d}, s{S_OOB}")
        E("s_nop 1")
        E(f"v_cndmask_b32_e32 v{d}, v{d}, v{TT(1)}, vcc")
    L(ldone)


def halo_loads(issue=True, setn=0):
    for i in range(3 if issue else 0):
        E(f"buffer_load_dwordx4 {vr(HSET(setn, i), 4)}, v{VHOFF[i]}, s[{S_INR}:{S_INR + 3}], s{S_LC64} offen")
    # lc = lc + 1 == nC ? 0 : lc + 1;  lp += lc == 0
    E(f"s_add_u32 s{S_LC}, s{S_LC}, 1")
    E(f"s_cmp_eq_u32 s{S_LC}, s{S_NC}")
    E(f"s_cselect_b32 s{S_LC}, 0, s{S_LC}")
    E(f"s_cmp_eq_u32 s{S_LC}, 0")
    E(f"s_addc_u32 s{S_LP}, s{S_LP}, 0")
    E(f"s_lshl_b32 s{S_LC64}, s{S_LC}, 6")


def halo_stores(setn=0):
    for i in range(3):
        E(f"ds_write_b128 v{VHST[i]}, {vr(HSET(setn, i), 4)}")


def raw_reads(jp, jj, mi):
    """the eight ds_read_b128 of step (jj, mi): returns instruction strings"""
    cx = (S2F if jp else 0) if jj == 0 else S1F
    cy = (S1F if jp else S2F) if jj == 0 else (S3F if jp else S2F)
    r = []
    for hf in range(2):
        base = mi * 10880 + hf * 16
        r.append(f"ds_read_b128 {vr(RAW(hf, 0), 4)}, v{VA} offset:{base + cx * 4}")
        r.append(f"ds_read_b128 {vr(RAW(hf, 1), 4)}, v{VB} offset:{base + cx * 4}")
        r.append(f"ds_read_b128 {vr(RAW(hf, 2), 4)}, v{VA} offset:{base + cy * 4}")
        r.append(f"ds_read_b128 {vr(RAW(hf, 3), 4)}, v{VB} offset:{base + cy * 4}")
    return r


def form_valu(jp, jj, slot, hf):
    """transform + three-way split of half hf of step (jj, .) into pc[slot]: 34 VALU, chains interleaved"""
    w = "-1.0" if not (jj == 1 and jp == 0) else "1.0"
    a, b, c, d = RAW(hf, 0), RAW(hf, 1), RAW(hf, 2), RAW(hf, 3)
    r = []
    if opt("slow_valu"):                  # the first version's forms (VOP3 / SGPR operands), kept for the A/B
        for e in range(4):
            r.append(f"v_fma_f32 v{a + e}, s{S_SGN}, v{b + e}, v{a + e}")       # qx = sgn * rb_x + ra_x
        for e in range(4):
            r.append(f"v_fma_f32 v{c + e}, s{S_SGN}, v{d + e}, v{c + e}")       # qy
        for e in range(4):
            r.append(f"v_fma_f32 v{a + e}, {w}, v{c + e}, v{a + e}")            # v = w * qy + qx
    else:                                 # the same values bit for bit: fma(sgn, b, a) as v_fmac; fma(+-1, qy, qx) = qx +- qy rounded once
        for e in range(4):
            r.append(f"v_fmac_f32_e32 v{a + e}, v{VSGN()}, v{b + e}")
        for e in range(4):
            r.append(f"v_fmac_f32_e32 v{c + e}, v{VSGN()}, v{d + e}")
        for e in range(4):
            r.append(f"v_{'add' if w == '1.0' else 'sub'}_f32_e32 v{a + e}, v{a + e}, v{c + e}")
    # pairs (0,1) -> piece dword hf*2, (2,3) -> hf*2+1;  temporaries: the b registers
    for piece in range(3):
        for p in range(2):
            r.append(f"v_perm_b32 v{PC(slot, piece) + hf * 2 + p}, v{a + 2 * p + 1}, v{a + 2 * p}, s{S_PERM}")
        if piece < 2:
            for e in range(4):
                r.append(f"v_and_b32_e32 v{b + e}, {'s' + str(S_MASK) if opt('slow_valu') else 'v' + str(VMASK())}, v{a + e}")
            for e in range(4):
                r.append(f"v_sub_f32_e32 v{a + e}, v{a + e}, v{b + e}")
    assert len(r) == 34
    return r


FIRST_CHUNK = [False]      # emitting the peeled first chunk of a patch: every accumulator chain starts from the constant 0
#                            (no clearing of the 128 / 64 accumulators per patch: they were 8 % of the VALU operations of a 64-channel layer)
def mfmas(jj, mi, slot):
    r = []
    for nt in range(2):
        acc = vr(ACC(jj, nt, mi), 16)
        for k, (pa, pb) in enumerate(((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))):
            src = "0" if (FIRST_CHUNK[0] and k == 0 and not opt("zero_acc")) else acc
            r.append(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(PC(slot, pa), 4)}, {vr(BX(jj, nt, pb), 4)}, {src}")
    return r


def weight_load(jj, nt, p):
    return f"buffer_load_dwordx4 {vr(BX(jj, nt, p), 4)}, v{VLANE16}, s[{S_UR}:{S_UR + 3}], s{S_WO[jj][nt]} offen offset:{p * 1024}"


STAMP_PAIRS = [60, 62, 64, 66, 70, 72, 74, 76]
def stamp(i):
    """timing-only (steptimes): s_memtime into pair i; placed only where no LDS read is in flight (lgkmcnt returns out of order with SMEM)"""
    if opt("steptimes"):
        E(f"s_memtime s[{STAMP_PAIRS[i]}:{STAMP_PAIRS[i] + 1}]")
        E("s_waitcnt lgkmcnt(0)")


def emit_step_spread(jp, s):
    """Step s with the raw-operand reads spread over the MFMA gaps (at most two ds_read_b128 per gap, none waited for in the
    step it was issued in): the raw registers are two halves of four 16-byte registers (input channels 0-3 / 4-7 of the lane's
    eight); half 0 of step s + 2 is requested in gaps 7-8 of step s (its registers are free once the first half of step s + 1's
    transform has run), half 1 of step s + 1 in gaps 0-1.  Reads of the next chunk begin in gap 7 of step 2: B1 sits in front
    of step 2 and the read bases flip to the other buffer inside it."""
    jj, mi, slot = s >> 1, s & 1, s & 1
    n1, n2 = (s + 1) & 3, (s + 2) & 3
    stamp((2, 3, 4, 6)[s])                # starts of steps 0, 1; arrival at B1; start of step 3
    if s == 2:
        E("s_waitcnt lgkmcnt(0)")
        if not opt("no_barrier"):
            E("s_barrier")                # B1: chunk c + 1 is complete in the other buffer
        stamp(5)
    if s == 0:
        # this component's weight pieces (requested in step 1 of the previous chunk).  VMEM operations retire in order: younger than
        # them are the other component's six pieces (step 3) and, when the staging stood in front of this step, this chunk's three
        # halo loads; with the staging inside the step they are issued BEHIND this wait.  The halo registers (older) are covered.
        E(f"s_waitcnt vmcnt({6 if opt('stage_in_step0', 1) else 9})")
    if s == 2:
        E("s_waitcnt vmcnt(9)")           # pieces requested in step 3 of the previous chunk; younger: 3 halo loads + 6 pieces of step 1
    # two wait states between the v_perm that wrote the last dword of this step's low-order operand piece (tail of the previous
    # step: only its final MFMA lies between) and the first MFMA, which reads that piece
    E("s_nop 1")
    mf = mfmas(jj, mi, slot)
    v0 = form_valu(jp, n1 >> 1, slot ^ 1, 0)
    v1 = form_valu(jp, n1 >> 1, slot ^ 1, 1)
    if opt("no_valu"):
        v0, v1 = [], []
    r1 = raw_reads(jp, n1 >> 1, n1 & 1)[4:8]        # half 1 of the next step
    r2 = raw_reads(jp, n2 >> 1, n2 & 1)[0:4]        # half 0 of the step after it
    if opt("no_ldsread"):
        r1, r2 = [], []
    wl = {}
    if mi == 1 and not opt("no_wload"):
        wl = {1: [weight_load(jj, 0, 2)], 4: [weight_load(jj, 0, 1)], 5: [weight_load(jj, 0, 0)],
              7: [weight_load(jj, 1, 2)], 10: [weight_load(jj, 1, 1)], 11: [weight_load(jj, 1, 0)]}
    def take(lst, n):
        for _ in range(min(n, len(lst))):
            E(lst.pop(0))
    extra = {}
    if s == 0 and opt("stage_in_step0", 1):
        # Parking chunk c + 1 (halo registers -> the idle buffer) and requesting chunk c + 2 ride in the gaps of step 0 instead of
        # standing in front of it: the matrix pipe starts right behind B0.  vmcnt(9) above covers the halo registers (they are
        # older than the weight pieces it waits for).
        if not opt("no_halo"):
            for i in range(3):
                extra.setdefault(i, []).append(f"ds_write_b128 v{VHST[i]}, {vr(HREG(i), 4)}")
            for i in range(3):
                extra.setdefault(3 + i, []).append(f"buffer_load_dwordx4 {vr(HREG(i), 4)}, v{VHOFF[i]}, s[{S_INR}:{S_INR + 3}], s{S_LC64} offen")
        extra.setdefault(6, []).extend([
            f"s_add_u32 s{S_LC}, s{S_LC}, 1", f"s_cmp_eq_u32 s{S_LC}, s{S_NC}", f"s_cselect_b32 s{S_LC}, 0, s{S_LC}",
            f"s_cmp_eq_u32 s{S_LC}, 0", f"s_addc_u32 s{S_LP}, s{S_LP}, 0", f"s_lshl_b32 s{S_LC64}, s{S_LC}, 6"])
        extra.setdefault(3, []).extend([f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}" for i in range(3)])
    a0, a1 = opt("sp_v0", (2, 6)), opt("sp_v1", (7, 10))      # gap ranges of the two transform halves
    per0 = -(-34 // (a0[1] - a0[0] + 1))
    per1 = -(-34 // (a1[1] - a1[0] + 1))
    pr = opt("prio")                      # timing experiments: s_setprio per (jp, step)
    if pr and pr[jp][s] != pr[jp][(s + 3) & 3]:
        E(f"s_setprio {pr[jp][s]}")
    nlds = 0                              # LDS operations issued in this step so far (all younger than half 0's reads)
    for k in range(12):
        if not opt("no_mfma"):
            E(mf[k])
        for x in wl.get(k, []):
            E(x)
        for x in extra.get(k, []):
            E(x)
            nlds += x.startswith("ds_")
        if k in (0, 1):
            nlds += min(2, len(r1))
            take(r1, 2)
        if k == a0[0]:
            # half 0 (requested in the previous step) complete: everything issued in this step so far may stay in flight
            E(f"s_waitcnt lgkmcnt({nlds})")
        if a0[0] <= k <= a0[1]:
            take(v0, per0)
        if k == a1[0]:
            assert not v0
            E("s_waitcnt lgkmcnt(0)")
            if s == 2:
                E(f"v_xor_b32_e32 v{VA}, 0x{BUFX:x}, v{VA}")
                E(f"v_xor_b32_e32 v{VB}, 0x{BUFX:x}, v{VB}")
        if k in (a1[0], a1[0] + 1):
            take(r2, 2)
        if a1[0] <= k <= a1[1]:
            take(v1, per1)
    assert not v1 and not r1 and not r2


def emit_step_pp(jp, s):
    """Ping-pong schedule (option pingpong; an experiment of record -- measured 1.8 % slower than the spread schedule, see
    profiles/r05_asm_experiments.txt -- and not maintained against later changes of the register map: the cold-code temporaries of
    setup_load / the epilogue would have to move out of the raw set, which it keeps in flight across chunk tops): a wave alternates a PURE matrix phase (the step's twelve MFMAs back to back) with a
    transform phase (wait for the raw operands requested before the burst, 68 VALU into the single operand-piece slot), and the two
    waves of a SIMD run the phases in opposite order -- waves 0-3 (jp 0): M(s) then V(s + 1); waves 4-7 (jp 1): V(s) then M(s) -- so
    one wave's transform always lies beside its partner's burst (tools/ubench: a burst of 8 MFMAs followed by 48 VALU, two waves per
    SIMD: both finish together at 100 % of the matrix pipe; the same work interleaved per gap: 75-93 %).  One operand-piece slot and
    one raw set per wave; the raw reads of the next transform are issued right in front of the burst."""
    jj, mi = s >> 1, s & 1
    X = jp == 0
    ns = (s + 1) & 3
    stamp(2 + s if s < 3 else 5)          # starts of steps 0..2; arrival at B1
    if s == 3:
        E("s_waitcnt lgkmcnt(0)")
        if not opt("no_barrier"):
            E("s_barrier")                # B1: the next chunk is complete in the other buffer (first read of it: this step)
        stamp(6)
    def reads(step):
        if not opt("no_ldsread"):
            for r in raw_reads(jp, step >> 1, step & 1):
                E(r)
    def flip():
        E(f"v_xor_b32_e32 v{VA}, 0x{BUFX:x}, v{VA}")
        E(f"v_xor_b32_e32 v{VB}, 0x{BUFX:x}, v{VB}")
    def staging():
        if opt("no_halo"):
            return
        for i in range(3):
            E(f"ds_write_b128 v{VHST[i]}, {vr(HREG(i), 4)}")
        for i in range(3):
            E(f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}")
        halo_loads(True)
    def V(step):                          # transform + split of `step`'s operands into slot 0
        E("s_waitcnt lgkmcnt(0)")
        if not opt("no_valu"):
            for hf in range(2):
                for x in form_valu(jp, step >> 1, 0, hf):
                    E(x)
    def M():
        mf = mfmas(jj, mi, 0)
        wl = {}
        if mi == 1 and not opt("no_wload"):
            wl = {1: [weight_load(jj, 0, 2)], 4: [weight_load(jj, 0, 1)], 5: [weight_load(jj, 0, 0)],
                  7: [weight_load(jj, 1, 2)], 10: [weight_load(jj, 1, 1)], 11: [weight_load(jj, 1, 0)]}
        if opt("pp_prio"):
            E("s_setprio 1")
        for k in range(12):
            if not opt("no_mfma"):
                E(mf[k])
            for x in wl.get(k, []):
                E(x)
        if opt("pp_prio"):
            E("s_setprio 0")
    if X:
        if s == 3:
            flip()
        reads(ns)                         # operands of the NEXT step: requested in front of the burst, transformed behind it
        if s == 0:
            E("s_waitcnt vmcnt(6)")       # weight pieces of component 0 (step 1 of the previous chunk); younger: the six of step 3
        if s == 2:
            E("s_waitcnt vmcnt(9)")       # pieces of step 3 of the previous chunk; younger: 3 halo loads + 6 pieces of step 1
        M()
        V(ns)
        if s == 0:
            staging()                     # halo registers are older than the pieces waited for above
    else:
        if s == 0:
            E("s_waitcnt vmcnt(12)")      # the halo registers (both sets of weight pieces are younger)
        V(s)
        if s == 0:
            staging()
        if s == 3:
            flip()
        reads(ns)
        E("s_nop 1")
        if s in (0, 2):
            E("s_waitcnt vmcnt(9)")       # this component's pieces; younger: the other six + three halo loads (step 0: just issued)
        M()


def emit_step(jp, s):
    if opt("pingpong"):
        return emit_step_pp(jp, s)
    if opt("spread", 1):
        return emit_step_spread(jp, s)
    jj, mi, slot = s >> 1, s & 1, s & 1
    ns = (s + 1) & 3
    njj, nmi = ns >> 1, ns & 1
    if s == 2:
        pass
    stamp(2 + s if s < 3 else 5)         # 2, 3, 4: starts of steps 0..2; 5: arrival at B1
    if s == opt("b1_step", 3):
        # B1: chunk c + 1 is complete in the other buffer (step 3's reads are the next chunk's step 0)
        E("s_waitcnt lgkmcnt(0)")
        if not opt("no_barrier"):
            E("s_barrier")
        stamp(6)
    if s == 3:
        E(f"v_xor_b32_e32 v{VA}, 0x{BUFX:x}, v{VA}")
        E(f"v_xor_b32_e32 v{VB}, 0x{BUFX:x}, v{VB}")
    if not opt("no_ldsread"):
        for r in raw_reads(jp, njj, nmi):
            E(r)
    if s in (0, 2):
        E("s_waitcnt vmcnt(9)")          # this component's weight pieces (requested a chunk ago)
    mf = mfmas(jj, mi, slot)
    v0 = form_valu(jp, njj, slot ^ 1, 0)
    v1 = form_valu(jp, njj, slot ^ 1, 1)
    wl = {}
    if mi == 1 and not opt("no_wload"):   # pieces of the next chunk into the registers this step has finished with
        wl = {1: [weight_load(jj, 0, 2)], 4: [weight_load(jj, 0, 1)], 5: [weight_load(jj, 0, 0)],
              7: [weight_load(jj, 1, 2)], 10: [weight_load(jj, 1, 1)], 11: [weight_load(jj, 1, 0)]}
    def take(lst, n):
        for _ in range(min(n, len(lst))):
            E(lst.pop(0))
    if opt("no_valu"):
        v0, v1 = [], []
    f0 = opt("valu_from", 2)                     # first gap with transform work
    n0 = 5 if f0 >= 2 else 6 - f0                # gaps of the first half
    per0 = -(-34 // n0)
    per1 = -(-34 // (10 - (f0 + n0 - 1)))
    for k in range(12):
        if not opt("no_mfma"):
            E(mf[k])
        for x in wl.get(k, []):
            E(x)
        if k == f0:
            E("s_waitcnt lgkmcnt(4)")
        if f0 <= k < f0 + n0:
            take(v0, per0)
        if k == f0 + n0 - 1:
            assert not v0
            E("s_waitcnt lgkmcnt(0)")
        if f0 + n0 <= k <= 10:
            take(v1, per1)
    assert not v1


def emit_chunk(jp):
    """one 16-channel chunk: B0, park chunk c + 1, request chunk c + 2, four steps"""
    lskip = newlabel("nosetup")
    E(f"s_cmp_lg_u32 s{S_LC}, 0")
    E(f"s_cbranch_scc1 {lskip}")
    E(f"s_cmp_ge_u32 s{S_LP}, s{S_NPATCH}")
    E(f"s_cbranch_scc1 {lskip}")
    setup_load()
    L(lskip)
    E("s_waitcnt lgkmcnt(0)")
    stamp(0)                              # arrival at B0
    if not opt("no_barrier"):
        E("s_barrier")                    # B0
    stamp(1)
    inl = (opt("spread", 1) and opt("stage_in_step0", 1)) or opt("pingpong")
    if not inl:
        E("s_waitcnt vmcnt(12)")          # the halo registers (two sets of weight pieces are younger)
        if not opt("no_halo"):
            halo_stores()
        halo_loads(not opt("no_halo"))
        for i in range(3):
            E(f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}")
    # weight soffsets of chunk cn = c + 1 == nC ? 0 : c + 1
    E(f"s_add_u32 s{S_T[0]}, s{S_C}, 1")
    E(f"s_cmp_eq_u32 s{S_T[0]}, s{S_NC}")
    E(f"s_cselect_b32 s{S_T[0]}, 0, s{S_T[0]}")
    E(f"s_mul_i32 s{S_WO[0][0]}, s{S_T[0]}, 0xc000")
    E(f"s_add_u32 s{S_WO[0][1]}, s{S_WO[0][0]}, s{S_NTSTRIDE}")
    E(f"s_add_u32 s{S_WO[1][0]}, s{S_WO[0][0]}, 0xc00")
    E(f"s_add_u32 s{S_WO[1][1]}, s{S_WO[0][1]}, 0xc00")
    for s in range(4):
        emit_step(jp, s)
    if opt("steptimes"):                  # stamp 7: end of the chunk; chunk 8's stamps -> out[(wg * 8 + wave) * 8 ..]
        E("s_waitcnt lgkmcnt(0)")
        E("s_memtime s[92:93]")
        E("s_waitcnt lgkmcnt(0)")
        lno = newlabel("nostore")
        E(f"s_cmp_lg_u32 s{S_C}, 8")
        E(f"s_cbranch_scc1 {lno}")
        E(f"s_mov_b32 s{S_OUTR}, s{S_OUT}")
        E(f"s_and_b32 s{S_OUTR + 1}, s{S_OUT + 1}, 0xffff")
        E(f"s_mov_b32 s{S_OUTR + 2}, 0x7ffffff0")
        E(f"s_mov_b32 s{S_OUTR + 3}, 0x00020000")
        E("s_lshl_b32 s94, s2, 3")
        E(f"s_lshl_b32 s95, s{S_JP}, 2")
        E(f"s_add_u32 s95, s95, s{S_WI}")
        E("s_add_u32 s94, s94, s95")
        E("s_lshl_b32 s94, s94, 5")
        E("s_mov_b64 s[96:97], exec")
        E(f"v_and_b32_e32 v{TT(0)}, 63, v{VTID}")
        E(f"v_cmp_eq_u32_e32 vcc, 0, v{TT(0)}")
        E("s_and_b64 exec, exec, vcc")
        for k in range(4):
            a, b = STAMP_PAIRS[2 * k], (STAMP_PAIRS[2 * k + 1] if k < 3 else 92)
            if k == 3:
                a = STAMP_PAIRS[6]
            E(f"v_mov_b32_e32 v{TT(0)}, s{a}")
            E(f"v_mov_b32_e32 v{TT(1)}, s{b}")
            E(f"v_mov_b32_e32 v{VHOFF[0]}, 0")
            E(f"buffer_store_dwordx2 v[{TT(0)}:{TT(1)}], v{VHOFF[0]}, s[{S_OUTR}:{S_OUTR + 3}], s94 offen offset:{8 * k}")
            E("s_waitcnt vmcnt(0)")
        E("s_mov_b64 exec, s[96:97]")
        L(lno)


def pk2(op, d, a, b, neg_a=False, neg_b=False):
    """two fp32 operations per lane in one VOP3P instruction on even-aligned register pairs (bit-identical to the scalar forms).
    Only for the epilogue: next to an MFMA in flight the packed forms wait for the matrix pipe (tools/ubench)."""
    assert d % 2 == 0 and a % 2 == 0 and b % 2 == 0
    mod = ""
    if neg_a or neg_b:
        mod = f" neg_lo:[{int(neg_a)},{int(neg_b)}] neg_hi:[{int(neg_a)},{int(neg_b)}]"
    return f"v_pk_{op}_f32 v[{d}:{d + 1}], v[{a}:{a + 1}], v[{b}:{b + 1}]{mod}"


def pkfma2(d, a, b, c):
    assert d % 2 == 0 and a % 2 == 0 and b % 2 == 0 and c % 2 == 0
    return f"v_pk_fma_f32 v[{d}:{d + 1}], v[{a}:{a + 1}], v[{b}:{b + 1}], v[{c}:{c + 1}]"


def emit_finish_math(Z, q, SCq, SHq):
    """z_i = share(jp 0) + share(jp 1); ya = (z0 + z1) + z2; yb = (z1 - z2) - z3; y = sc * y + sh -- the scalar sequence's values, packed"""
    for i in range(4):
        for e in (0, 2):
            E(pk2("add", Z(q, 0, i) + e, Z(q, 0, i) + e, Z(q, 1, i) + e))
    ya, yb = Z(q, 1, 0), Z(q, 1, 1)
    for e in (0, 2):
        E(pk2("add", ya + e, Z(q, 0, 0) + e, Z(q, 0, 1) + e))
    for e in (0, 2):
        E(pk2("add", ya + e, ya + e, Z(q, 0, 2) + e))
    for e in (0, 2):
        E(pk2("add", yb + e, Z(q, 0, 1) + e, Z(q, 0, 2) + e, neg_b=True))
    for e in (0, 2):
        E(pk2("add", yb + e, yb + e, Z(q, 0, 3) + e, neg_b=True))
    for y in (ya, yb):
        for e in (0, 2):
            E(pkfma2(y + e, SCq + e, y + e, SHq + e))


def emit_epilogue(jp):
    # free registers: the second operand slot (v188..v199) and raw half 1 (v216..v231); raw half 0 holds the next patch's
    # first reads (spread schedule) and stays untouched
    E0 = 188
    CQ, VT = 224, 225
    VZ0, VZ1, VOUT, VPOOL = E0 + 0, E0 + 1, E0 + 2, E0 + 3
    e0, e1, e2, e3 = E0 + 8, E0 + 9, E0 + 10, E0 + 11
    TMP = [216 + i for i in range(4)] * 2
    # per-channel scale / shift of the FINISHING unit's channel quad, both n tiles (reader side: two of the four shares are then plain
    # copies of the accumulators and the other two one add / subtract -- 32 instead of 96 VALU per n tile in the write phase, 16 fma in
    # the finishing pass); e0..e3 are dead once the addresses are formed
    SC4 = [E0 + 4, 226]
    SH4 = [e0, 220]
    assert not opt("pingpong")
    E("s_nop 7")
    E("s_nop 7")
    E("s_nop 7")
    E(f"s_add_u32 s{S_P}, s{S_PBEGIN}, s{S_PI}")
    patch_coords(S_P)
    # output / pooled descriptors of the image
    E(f"s_mul_i32 s{S_T[6]}, s{S_IMG}, s{S_OUTIMGB}")
    E(f"s_mul_hi_u32 s{S_T[7]}, s{S_IMG}, s{S_OUTIMGB}")
    E(f"s_add_u32 s{S_OUTR}, s{S_OUT}, s{S_T[6]}")
    E(f"s_addc_u32 s{S_OUTR + 1}, s{S_OUT + 1}, s{S_T[7]}")
    E(f"s_and_b32 s{S_OUTR + 1}, s{S_OUTR + 1}, 0xffff")
    # pooled image bytes = (H/2) * (W/2) * ldpool * 4
    E(f"s_lshr_b32 s{S_T[4]}, s{S_H}, 1")
    E(f"s_lshr_b32 s{S_T[5]}, s{S_W}, 1")
    E(f"s_mul_i32 s{S_T[4]}, s{S_T[4]}, s{S_T[5]}")
    E(f"s_mul_i32 s{S_T[4]}, s{S_T[4]}, s{S_LDPOOL}")
    E(f"s_lshl_b32 s{S_T[4]}, s{S_T[4]}, 2")
    E(f"s_mov_b32 s{S_POOLR + 2}, s{S_T[4]}")
    E(f"s_mul_i32 s{S_T[6]}, s{S_IMG}, s{S_T[4]}")
    E(f"s_mul_hi_u32 s{S_T[7]}, s{S_IMG}, s{S_T[4]}")
    E(f"s_add_u32 s{S_POOLR}, s{S_POOL}, s{S_T[6]}")
    E(f"s_addc_u32 s{S_POOLR + 1}, s{S_POOL + 1}, s{S_T[7]}")
    E(f"s_and_b32 s{S_POOLR + 1}, s{S_POOLR + 1}, 0xffff")
    # finishing unit of the thread
    E(f"v_and_b32_e32 v{CQ}, 7, v{VTID}")
    E(f"v_lshrrev_b32_e32 v{VT}, 3, v{VTID}")
    E(f"v_and_b32_e32 v{e0}, 32, v{VT}")
    E(f"v_and_b32_e32 v{e1}, 3, v{VT}")
    E(f"v_bfe_u32 v{e2}, v{VT}, 3, 2")                       # (T & 31) >> 3
    E(f"v_lshl_add_u32 v{e1}, v{e2}, 2, v{e1}")
    E(f"v_lshl_add_u32 v{e0}, v{e1}, 1, v{e0}")
    E(f"v_bfe_u32 v{e2}, v{VT}, 2, 1")
    E(f"v_add_u32_e32 v{e0}, v{e0}, v{e2}")                  # Tslot
    E(f"v_lshlrev_b32_e32 v{e0}, 7, v{e0}")
    E(f"v_lshl_add_u32 v{VZ0}, v{CQ}, 4, v{e0}")
    E(f"v_add_u32_e32 v{VZ1}, 0x10000, v{VZ0}")
    E(f"v_lshrrev_b32_e32 v{e1}, 4, v{VT}")
    E(f"v_lshl_add_u32 v{e1}, v{e1}, 1, s{S_Y0}")            # oy
    E(f"v_and_b32_e32 v{e2}, 15, v{VT}")
    E(f"v_lshl_add_u32 v{e2}, v{e2}, 1, s{S_X0}")            # ox
    E(f"v_mad_u32_u24 v{e3}, v{e1}, s{S_W}, v{e2}")
    E(f"v_mul_lo_u32 v{e3}, v{e3}, s{S_LDOUT}")
    E(f"v_lshl_add_u32 v{e3}, v{CQ}, 2, v{e3}")
    E(f"v_lshlrev_b32_e32 v{VOUT}, 2, v{e3}")
    E(f"v_add_u32_e32 v{VOUT}, s{S_N64X4}, v{VOUT}")
    # pooled pixel
    E(f"v_lshrrev_b32_e32 v{e1}, 1, v{e1}")
    E(f"v_lshrrev_b32_e32 v{e2}, 1, v{e2}")
    E(f"v_mad_u32_u24 v{e3}, v{e1}, s{S_T[5]}, v{e2}")
    E(f"v_mul_lo_u32 v{e3}, v{e3}, s{S_LDPOOL}")
    E(f"v_lshl_add_u32 v{e3}, v{CQ}, 2, v{e3}")
    E(f"v_lshlrev_b32_e32 v{VPOOL}, 2, v{e3}")
    E(f"v_add_u32_e32 v{VPOOL}, s{S_N64X4}, v{VPOOL}")
    # scale / shift quads (n0 = nblock * 64 + nt * 32 + cq * 4): requested first, used last; a missing array is 1 / 0
    E(f"v_lshlrev_b32_e32 v{CQ}, 4, v{CQ}")                   # (cq is not needed any more)
    for (ptr, rs, regs, dflt) in ((S_SCALE, S_SCR, SC4, "1.0"), (S_SHIFT, S_SHR, SH4, "0")):
        ln, ldn = newlabel("nul"), newlabel("nud")
        E(f"s_cmp_eq_u64 s[{ptr}:{ptr + 1}], 0")
        E(f"s_cbranch_scc1 {ln}")
        for nt in range(2):
            E(f"buffer_load_dwordx4 {vr(regs[nt], 4)}, v{CQ}, s[{rs}:{rs + 3}], s{S_N64X4} offen offset:{nt * 128}")
        E(f"s_branch {ldn}")
        L(ln)
        for nt in range(2):
            for e in range(4):
                E(f"v_mov_b32_e32 v{regs[nt] + e}, {dflt}")
        L(ldn)
    # add-TID bases of this wave's two shares
    #   q = 0: jp 0 -> region (0,0) = slot 0 (the consumed raw buffer, even chunk count), jp 1 -> region (0,1) = slot 1
    #   q = 1: slot 2 + jp, biased by ZBIAS
    E(f"s_lshl_b32 s{S_T[0]}, s{S_WI}, 13")                  # wi * 64 * 32 * 4
    if jp:
        E(f"s_add_u32 s{S_T[0]}, s{S_T[0]}, 0x{RAWB:x}")
    E(f"s_lshl_b32 s{S_T[1]}, s{S_WI}, 13")
    E(f"s_add_u32 s{S_T[1]}, s{S_T[1]}, 0x{(2 + jp) * RAWB - ZBIAS:x}")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")                       # every wave has finished reading the consumed raw buffer
    # (no vmcnt wait here: the write phase reads accumulators only; the scale / shift quads requested above are first needed by the
    # finishing pass of n tile 0, a whole write phase and a barrier later)
    for nt in range(2):
        def share2(q, mi, r, t):         # -> the registers of the shares of accumulators r, r + 1;  jp 0: q0 = m0 + m1, q1 = m1;  jp 1: q0 = m0, q1 = -m0 - m1
            m0, m1 = ACC(0, nt, mi) + r, ACC(1, nt, mi) + r
            if jp == 0 and q == 0:
                E(pk2("add", t, m0, m1))
                return [t, t + 1]
            if jp == 0:
                return [m1, m1 + 1]
            if q == 0:
                return [m0, m0 + 1]
            E(pk2("add", t, m0, m1, neg_a=True, neg_b=True))
            return [t, t + 1]
        for q in range(2):
            E(f"s_mov_b32 m0, s{S_T[q]}")
            E("s_nop 0")
            g = 0
            for mi in range(2):
                for r0 in range(0, 16, 4):
                    ts = TMP[(g & 1) * 4:(g & 1) * 4 + 4]
                    g += 1
                    src = share2(q, mi, r0, ts[0]) + share2(q, mi, r0 + 2, ts[2])
                    E("s_nop 0")
                    for k in range(4):
                        off = (32 * mi + 2 * (r0 + k)) * 128 + (ZBIAS if q else 0)
                        assert "noaddtid" not in DEBUG
                        E(f"ds_write_addtid_b32 v{src[k]} offset:{off}")
        E("s_waitcnt lgkmcnt(0)")
        E("s_barrier")
        # finishing pass of unit (T, cq): 16 share reads into the dead accumulators of this n tile
        zb = [ACC(0, nt, 0), ACC(0, nt, 1), ACC(1, nt, 0), ACC(1, nt, 1)]
        def Z(q, j, i):
            k = (q * 2 + j) * 4 + i
            return zb[k // 4] + (k % 4) * 4
        for q in range(2):
            for j in range(2):
                for i in range(4):
                    E(f"ds_read_b128 {vr(Z(q, j, i), 4)}, v{VZ1 if q else VZ0} offset:{j * RAWB + i * 8192}")
        # the q = 0 half of the reads first (LDS operations of a wave return in order): its arithmetic runs while the q = 1 half lands
        E("s_waitcnt vmcnt(0) lgkmcnt(8)" if nt == 0 else "s_waitcnt lgkmcnt(8)")
        emit_finish_math(Z, 0, SC4[nt], SH4[nt])
        E("s_waitcnt lgkmcnt(0)")
        emit_finish_math(Z, 1, SC4[nt], SH4[nt])
        lnr = newlabel("norelu")
        E(f"s_cmp_eq_u32 s{S_RELU}, 0")
        E(f"s_cbranch_scc1 {lnr}")
        for q in range(2):
            for y in (Z(q, 1, 0), Z(q, 1, 1)):
                for e in range(4):
                    E(f"v_max_f32_e32 v{y + e}, 0, v{y + e}")
        L(lnr)
        ya0, yb0, ya1, yb1 = Z(0, 1, 0), Z(0, 1, 1), Z(1, 1, 0), Z(1, 1, 1)
        E(f"buffer_store_dwordx4 {vr(ya0, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], 0 offen offset:{nt * 128} nt")
        E(f"buffer_store_dwordx4 {vr(ya1, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], s{S_LD4} offen offset:{nt * 128} nt")
        E(f"buffer_store_dwordx4 {vr(yb0, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], s{S_SW4} offen offset:{nt * 128} nt")
        E(f"buffer_store_dwordx4 {vr(yb1, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], s{S_SUMOFF} offen offset:{nt * 128} nt")
        lnp = newlabel("nopool")
        E(f"s_cmp_eq_u64 s[{S_POOL}:{S_POOL + 1}], 0")
        E(f"s_cbranch_scc1 {lnp}")
        pm = Z(0, 0, 0)
        for e in range(4):
            E(f"v_max_f32_e32 v{pm + e}, v{ya0 + e}, v{yb0 + e}")
        for e in range(4):
            E(f"v_max_f32_e32 v{Z(0, 0, 1) + e}, v{ya1 + e}, v{yb1 + e}")
        for e in range(4):
            E(f"v_max_f32_e32 v{pm + e}, v{pm + e}, v{Z(0, 0, 1) + e}")
        E(f"buffer_store_dwordx4 {vr(pm, 4)}, v{VPOOL}, s[{S_POOLR}:{S_POOLR + 3}], 0 offen offset:{nt * 128}")
        L(lnp)
        # (the accumulators -- registers of the finishing pass -- are NOT cleared: the first chunk of a patch is a peeled copy of the
        # chunk loop whose accumulator chains start from the constant 0; option zero_acc restores the clearing for the A/B)
        if opt("zero_acc"):
            E("s_nop 1")
            for b in sorted(zb):
                for r in range(16):
                    E(f"v_mov_b32_e32 v{b + r}, 0")
        E("s_waitcnt lgkmcnt(0)")
        E("s_barrier")                   # the regions are rewritten by the next pass / receive the next raw chunk


# =====================================================================================================================
# Narrow kernels (32 output channels per workgroup; asm form of wino3x3_cp_kernel<1, false, true, *>): a chunk is only six MFMAs
# per wave and step, so the loop is bound by the transform's VALU issue and by latency, not by the matrix pipe.  Against the wide
# kernel: (i) the wave's weight pieces of ALL chunks stay in registers (2 or 4 chunks: 48 / 96 registers; the accumulators are only
# 64) -- no weight stream at all; (ii) the halo is requested two chunks ahead through two register sets whose parity is static
# (the chunk loop is unrolled); (iii) scale / shift are applied by the finishing pass (two of the four shares are plain copies of
# the accumulators).  Same arithmetic order as the C++ kernel: bitwise equal outputs.
# =====================================================================================================================
def mfmas_n(jj, mi, slot, c):
    acc = vr(ACC(jj, 0, mi), 16)
    return [f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(PC(slot, pa), 4)}, {vr(WN(c, jj, pb), 4)}, {'0' if (c == 0 and k == 0) else acc}"
            for k, (pa, pb) in enumerate(((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)))]


def emit_step_n(jp, s, c):
    jj, mi, slot = s >> 1, s & 1, s & 1
    n1, n2 = (s + 1) & 3, (s + 2) & 3
    if s == 2:
        E("s_waitcnt lgkmcnt(0)")
        E("s_barrier")                    # B1
    E("s_nop 1")                          # the previous step ends with the v_perm that writes this step's low-order operand piece, and the
    #                                       first MFMA reads that piece: two wait states between a VALU write and an MFMA read of it
    mf = mfmas_n(jj, mi, slot, c)
    v0 = form_valu(jp, n1 >> 1, slot ^ 1, 0)
    v1 = form_valu(jp, n1 >> 1, slot ^ 1, 1)
    r1 = raw_reads(jp, n1 >> 1, n1 & 1)[4:8]
    r2 = raw_reads(jp, n2 >> 1, n2 & 1)[0:4]
    extra = {}
    if s == 0:
        hs = (c + 1) & 1                  # the set holding chunk c + 1; it is refilled with chunk c + 3
        E("s_waitcnt vmcnt(3)")           # (only the three loads of chunk c + 2 are younger)
        for i in range(3):
            extra.setdefault(i, []).append(f"ds_write_b128 v{VHST[i]}, {vr(HSET(hs, i), 4)}")
        extra.setdefault(3, []).extend([f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}" for i in range(3)])
        for i in range(3):
            extra.setdefault(3 + i, []).append(f"buffer_load_dwordx4 {vr(HSET(hs, i), 4)}, v{VHOFF[i]}, s[{S_INR}:{S_INR + 3}], s{S_LC64} offen")
        extra.setdefault(5, []).extend([
            f"s_add_u32 s{S_LC}, s{S_LC}, 1", f"s_cmp_eq_u32 s{S_LC}, s{S_NC}", f"s_cselect_b32 s{S_LC}, 0, s{S_LC}",
            f"s_cmp_eq_u32 s{S_LC}, 0", f"s_addc_u32 s{S_LP}, s{S_LP}, 0", f"s_lshl_b32 s{S_LC64}, s{S_LC}, 6"])
    def take(lst, n):
        for _ in range(min(n, len(lst))):
            E(lst.pop(0))
    nlds = 0
    for k in range(6):
        E(mf[k])
        for x in extra.get(k, []):
            E(x)
            nlds += x.startswith("ds_")
        if k in (0, 1):
            take(r1, 2)
            nlds += 2
        if k == 1:
            E(f"s_waitcnt lgkmcnt({nlds})")   # half 0 (requested in the previous step); everything issued in this step may be in flight
        if k in (1, 2):
            take(v0, 17)
        if k == 3:
            E("s_waitcnt lgkmcnt(0)")
            if s == 2:
                E(f"v_xor_b32_e32 v{VA}, 0x{BUFX:x}, v{VA}")
                E(f"v_xor_b32_e32 v{VB}, 0x{BUFX:x}, v{VB}")
        if k in (3, 4):
            take(r2, 2)
        if k >= 3:
            take(v1, 12)
    assert not v0 and not v1 and not r1 and not r2


def emit_chunk_n(jp, c):
    lskip = newlabel("nosetup")
    E(f"s_cmp_lg_u32 s{S_LC}, 0")
    E(f"s_cbranch_scc1 {lskip}")
    E(f"s_cmp_ge_u32 s{S_LP}, s{S_NPATCH}")
    E(f"s_cbranch_scc1 {lskip}")
    setup_load()
    L(lskip)
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")                        # B0
    for s in range(4):
        emit_step_n(jp, s, c)


def emit_epilogue_n(jp):
    CQ, VT = 228, 229
    VZ0, VZ1, VOUT, VPOOL = 188, 189, 190, 191
    e0, e1, e2, e3 = 192, 193, 194, 195
    SC4, SH4 = 196, 216
    TMP = [220 + i for i in range(8)]
    E("s_nop 7")
    E("s_nop 7")
    E("s_nop 7")
    E(f"s_add_u32 s{S_P}, s{S_PBEGIN}, s{S_PI}")
    patch_coords(S_P)
    E(f"s_mul_i32 s{S_T[6]}, s{S_IMG}, s{S_OUTIMGB}")
    E(f"s_mul_hi_u32 s{S_T[7]}, s{S_IMG}, s{S_OUTIMGB}")
    E(f"s_add_u32 s{S_OUTR}, s{S_OUT}, s{S_T[6]}")
    E(f"s_addc_u32 s{S_OUTR + 1}, s{S_OUT + 1}, s{S_T[7]}")
    E(f"s_and_b32 s{S_OUTR + 1}, s{S_OUTR + 1}, 0xffff")
    E(f"s_lshr_b32 s{S_T[4]}, s{S_H}, 1")
    E(f"s_lshr_b32 s{S_T[5]}, s{S_W}, 1")
    E(f"s_mul_i32 s{S_T[4]}, s{S_T[4]}, s{S_T[5]}")
    E(f"s_mul_i32 s{S_T[4]}, s{S_T[4]}, s{S_LDPOOL}")
    E(f"s_lshl_b32 s{S_T[4]}, s{S_T[4]}, 2")
    E(f"s_mov_b32 s{S_POOLR + 2}, s{S_T[4]}")
    E(f"s_mul_i32 s{S_T[6]}, s{S_IMG}, s{S_T[4]}")
    E(f"s_mul_hi_u32 s{S_T[7]}, s{S_IMG}, s{S_T[4]}")
    E(f"s_add_u32 s{S_POOLR}, s{S_POOL}, s{S_T[6]}")
    E(f"s_addc_u32 s{S_POOLR + 1}, s{S_POOL + 1}, s{S_T[7]}")
    E(f"s_and_b32 s{S_POOLR + 1}, s{S_POOLR + 1}, 0xffff")
    E(f"v_and_b32_e32 v{CQ}, 7, v{VTID}")
    E(f"v_lshrrev_b32_e32 v{VT}, 3, v{VTID}")
    # per-channel scale / shift of the finishing unit's channel quad (n0 = nblock * 32 + cq * 4): requested first, used last
    E(f"v_lshlrev_b32_e32 v{e0}, 4, v{CQ}")
    for (ptr, rs, reg, dflt) in ((S_SCALE, S_SCR, SC4, "1.0"), (S_SHIFT, S_SHR, SH4, "0")):
        ln, ldn = newlabel("nul"), newlabel("nud")
        E(f"s_cmp_eq_u64 s[{ptr}:{ptr + 1}], 0")
        E(f"s_cbranch_scc1 {ln}")
        E(f"buffer_load_dwordx4 {vr(reg, 4)}, v{e0}, s[{rs}:{rs + 3}], s{S_N64X4} offen")
        E(f"s_branch {ldn}")
        L(ln)
        for e in range(4):
            E(f"v_mov_b32_e32 v{reg + e}, {dflt}")
        L(ldn)
    E(f"v_and_b32_e32 v{e0}, 32, v{VT}")
    E(f"v_and_b32_e32 v{e1}, 3, v{VT}")
    E(f"v_bfe_u32 v{e2}, v{VT}, 3, 2")
    E(f"v_lshl_add_u32 v{e1}, v{e2}, 2, v{e1}")
    E(f"v_lshl_add_u32 v{e0}, v{e1}, 1, v{e0}")
    E(f"v_bfe_u32 v{e2}, v{VT}, 2, 1")
    E(f"v_add_u32_e32 v{e0}, v{e0}, v{e2}")
    E(f"v_lshlrev_b32_e32 v{e0}, 7, v{e0}")
    E(f"v_lshl_add_u32 v{VZ0}, v{CQ}, 4, v{e0}")
    E(f"v_add_u32_e32 v{VZ1}, 0x10000, v{VZ0}")
    E(f"v_lshrrev_b32_e32 v{e1}, 4, v{VT}")
    E(f"v_lshl_add_u32 v{e1}, v{e1}, 1, s{S_Y0}")
    E(f"v_and_b32_e32 v{e2}, 15, v{VT}")
    E(f"v_lshl_add_u32 v{e2}, v{e2}, 1, s{S_X0}")
    E(f"v_mad_u32_u24 v{e3}, v{e1}, s{S_W}, v{e2}")
    E(f"v_mul_lo_u32 v{e3}, v{e3}, s{S_LDOUT}")
    E(f"v_lshl_add_u32 v{e3}, v{CQ}, 2, v{e3}")
    E(f"v_lshlrev_b32_e32 v{VOUT}, 2, v{e3}")
    E(f"v_add_u32_e32 v{VOUT}, s{S_N64X4}, v{VOUT}")
    E(f"v_lshrrev_b32_e32 v{e1}, 1, v{e1}")
    E(f"v_lshrrev_b32_e32 v{e2}, 1, v{e2}")
    E(f"v_mad_u32_u24 v{e3}, v{e1}, s{S_T[5]}, v{e2}")
    E(f"v_mul_lo_u32 v{e3}, v{e3}, s{S_LDPOOL}")
    E(f"v_lshl_add_u32 v{e3}, v{CQ}, 2, v{e3}")
    E(f"v_lshlrev_b32_e32 v{VPOOL}, 2, v{e3}")
    E(f"v_add_u32_e32 v{VPOOL}, s{S_N64X4}, v{VPOOL}")
    E(f"s_lshl_b32 s{S_T[0]}, s{S_WI}, 13")
    if jp:
        E(f"s_add_u32 s{S_T[0]}, s{S_T[0]}, 0x{RAWB:x}")
    E(f"s_lshl_b32 s{S_T[1]}, s{S_WI}, 13")
    E(f"s_add_u32 s{S_T[1]}, s{S_T[1]}, 0x{(2 + jp) * RAWB - ZBIAS:x}")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    # shares: jp 0: q0 = m0 + m1, q1 = m1 (the accumulator itself);  jp 1: q0 = m0 (itself), q1 = -m0 - m1
    for q in range(2):
        E(f"s_mov_b32 m0, s{S_T[q]}")
        E("s_nop 0")
        g = 0
        for mi in range(2):
            for r0 in range(0, 16, 4):
                ts = TMP[(g & 1) * 4:(g & 1) * 4 + 4]
                g += 1
                src = []
                for k in (0, 2):
                    m0, m1 = ACC(0, 0, mi) + r0 + k, ACC(1, 0, mi) + r0 + k
                    if jp == 0 and q == 0:
                        E(pk2("add", ts[k], m0, m1))
                        src += [ts[k], ts[k] + 1]
                    elif jp == 0:
                        src += [m1, m1 + 1]
                    elif q == 0:
                        src += [m0, m0 + 1]
                    else:
                        E(pk2("add", ts[k], m0, m1, neg_a=True, neg_b=True))
                        src += [ts[k], ts[k] + 1]
                E("s_nop 0")
                for k in range(4):
                    off = (32 * mi + 2 * (r0 + k)) * 128 + (ZBIAS if q else 0)
                    E(f"ds_write_addtid_b32 v{src[k]} offset:{off}")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")
    zb = [ACC(0, 0, 0), ACC(0, 0, 1), ACC(1, 0, 0), ACC(1, 0, 1)]
    def Z(q, j, i):
        k = (q * 2 + j) * 4 + i
        return zb[k // 4] + (k % 4) * 4
    for q in range(2):
        for j in range(2):
            for i in range(4):
                E(f"ds_read_b128 {vr(Z(q, j, i), 4)}, v{VZ1 if q else VZ0} offset:{j * RAWB + i * 8192}")
    E("s_waitcnt vmcnt(0) lgkmcnt(8)")   # scale / shift, and the q = 0 half of the reads; its arithmetic runs while the q = 1 half lands
    emit_finish_math(Z, 0, SC4, SH4)
    E("s_waitcnt lgkmcnt(0)")
    emit_finish_math(Z, 1, SC4, SH4)
    lnr = newlabel("norelu")
    E(f"s_cmp_eq_u32 s{S_RELU}, 0")
    E(f"s_cbranch_scc1 {lnr}")
    for q in range(2):
        for y in (Z(q, 1, 0), Z(q, 1, 1)):
            for e in range(4):
                E(f"v_max_f32_e32 v{y + e}, 0, v{y + e}")
    L(lnr)
    ya0, yb0, ya1, yb1 = Z(0, 1, 0), Z(0, 1, 1), Z(1, 1, 0), Z(1, 1, 1)
    E(f"buffer_store_dwordx4 {vr(ya0, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], 0 offen nt")
    E(f"buffer_store_dwordx4 {vr(ya1, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], s{S_LD4} offen nt")
    E(f"buffer_store_dwordx4 {vr(yb0, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], s{S_SW4} offen nt")
    E(f"buffer_store_dwordx4 {vr(yb1, 4)}, v{VOUT}, s[{S_OUTR}:{S_OUTR + 3}], s{S_SUMOFF} offen nt")
    lnp = newlabel("nopool")
    E(f"s_cmp_eq_u64 s[{S_POOL}:{S_POOL + 1}], 0")
    E(f"s_cbranch_scc1 {lnp}")
    pm = Z(0, 0, 0)
    for e in range(4):
        E(f"v_max_f32_e32 v{pm + e}, v{ya0 + e}, v{yb0 + e}")
    for e in range(4):
        E(f"v_max_f32_e32 v{Z(0, 0, 1) + e}, v{ya1 + e}, v{yb1 + e}")
    for e in range(4):
        E(f"v_max_f32_e32 v{pm + e}, v{pm + e}, v{Z(0, 0, 1) + e}")
    E(f"buffer_store_dwordx4 {vr(pm, 4)}, v{VPOOL}, s[{S_POOLR}:{S_POOLR + 3}], 0 offen")
    L(lnp)
    E("s_waitcnt lgkmcnt(0)")            # (no clearing of the accumulators: chunk 0's chains start from the constant 0)
    E("s_barrier")


def emit_patch_loop_n(jp):
    lp = newlabel("patch")
    L(lp)
    for c in range(CFG["nc"]):
        emit_chunk_n(jp, c)
    emit_epilogue_n(jp)
    E(f"s_add_u32 s{S_PI}, s{S_PI}, 1")
    E(f"s_cmp_lt_u32 s{S_PI}, s{S_NPATCH}")
    E(f"s_cbranch_scc1 {lp}")
    E(f"s_branch {END_LABEL}")


def emit_patch_loop(jp):
    lp, lc = newlabel("patch"), newlabel("chunk")
    L(lp)
    E(f"s_mov_b32 s{S_C}, 0")
    if not opt("zero_acc"):              # chunk 0, peeled (a layer has at least two chunks: Cp % 32 == 0)
        FIRST_CHUNK[0] = True
        emit_chunk(jp)
        FIRST_CHUNK[0] = False
        E(f"s_mov_b32 s{S_C}, 1")
    L(lc)
    emit_chunk(jp)
    E(f"s_add_u32 s{S_C}, s{S_C}, 1")
    E(f"s_cmp_lt_u32 s{S_C}, s{S_NC}")
    E(f"s_cbranch_scc1 {lc}")
    if opt("no_epilogue"):
        E("s_nop 7")
        E("s_nop 7")
        for r in range(128):
            E(f"v_mov_b32_e32 v{r}, 0")
    else:
        emit_epilogue(jp)
    E(f"s_add_u32 s{S_PI}, s{S_PI}, 1")
    E(f"s_cmp_lt_u32 s{S_PI}, s{S_NPATCH}")
    E(f"s_cbranch_scc1 {lp}")
    if opt("stamp"):                 # (cycles, 100 MHz ticks) of the workgroup's life -> out[2 * wg .. 2 * wg + 1] (wave 0 writes)
        E("s_waitcnt vmcnt(0) lgkmcnt(0)")
        E("s_memtime s[60:61]")
        E("s_memrealtime s[62:63]")
        E("s_waitcnt lgkmcnt(0)")
        E(f"s_sub_u32 s60, s60, s{S_PAD}")
        E(f"s_sub_u32 s62, s62, s{S_W1}")
        E("v_mov_b32_e32 v0, s60")
        E("v_mov_b32_e32 v1, s62")
        E(f"s_mov_b32 s{S_OUTR}, s{S_OUT}")
        E(f"s_and_b32 s{S_OUTR + 1}, s{S_OUT + 1}, 0xffff")
        E(f"s_mov_b32 s{S_OUTR + 2}, 0x7ffffff0")
        E(f"s_mov_b32 s{S_OUTR + 3}, 0x00020000")
        E("s_lshl_b32 s61, s2, 3")
        E(f"v_cmp_eq_u32_e32 vcc, 0, v{VTID}")
        E("s_and_saveexec_b64 s[64:65], vcc")
        E("v_mov_b32_e32 v2, 0")
        E(f"buffer_store_dwordx2 v[0:1], v2, s[{S_OUTR}:{S_OUTR + 3}], s61 offen")
        E("s_waitcnt vmcnt(0)")
        E("s_mov_b64 exec, s[64:65]")
    E(f"s_branch {END_LABEL}")


def emit_prologue():
    E("s_load_dwordx16 s[4:19], s[0:1], 0x0")
    E("s_load_dwordx8 s[20:27], s[0:1], 0x40")
    E("s_load_dwordx4 s[28:31], s[0:1], 0x60")
    E("s_load_dwordx2 s[32:33], s[0:1], 0x70")
    E(f"v_mov_b32_e32 v{VTID}, v0")
    E("s_waitcnt lgkmcnt(0)")
    if opt("stamp"):                 # timing-only: shader clock and 100 MHz real-time counter at the start of the workgroup
        E("s_memtime s[60:61]")
        E("s_memrealtime s[62:63]")
        E("s_waitcnt lgkmcnt(0)")
        E(f"s_mov_b32 s{S_PAD}, s60")
        E(f"s_mov_b32 s{S_W1}, s62")
    # item = (wg & 7) * per_xcd + (wg >> 3)
    t = S_T
    E(f"s_and_b32 s{t[0]}, s2, 7")
    E(f"s_mul_i32 s{t[0]}, s{t[0]}, s{S_PERXCD}")
    E(f"s_lshr_b32 s{t[1]}, s2, 3")
    E(f"s_add_u32 s{t[0]}, s{t[0]}, s{t[1]}")
    E(f"s_cmp_ge_u32 s{t[0]}, s{S_NITEMS}")
    E(f"s_cbranch_scc1 {END_LABEL}")
    divmod_magic(t[0], S_NGROUPS, S_MGNG, S_NBLOCK, t[1], t[2], t[3])
    E(f"s_mul_i32 s{S_PBEGIN}, s{t[1]}, s{S_PPB}")
    E(f"s_sub_i32 s{t[2]}, s{S_TOTAL}, s{S_PBEGIN}")
    E(f"s_min_i32 s{S_NPATCH}, s{S_PPB}, s{t[2]}")
    E(f"s_cmp_lt_i32 s{S_NPATCH}, 1")
    E(f"s_cbranch_scc1 {END_LABEL}")
    # constants
    E(f"s_mov_b32 s{S_MASK}, 0xffff0000")
    E(f"s_mov_b32 s{S_PERM}, 0x07060302")
    E(f"s_mov_b32 s{S_OOB}, 0x7fff0000")
    E(f"s_mul_i32 s{S_TXTY}, s{S_TX}, s{S_TY}")
    E(f"s_mul_i32 s{S_IMGB}, s{S_H}, s{S_W}")
    E(f"s_mul_i32 s{S_OUTIMGB}, s{S_IMGB}, s{S_LDOUT}")
    E(f"s_lshl_b32 s{S_OUTIMGB}, s{S_OUTIMGB}, 2")
    E(f"s_mul_i32 s{S_IMGB}, s{S_IMGB}, s{S_LDIN}")
    E(f"s_lshl_b32 s{S_IMGB}, s{S_IMGB}, 2")
    E(f"s_mul_i32 s{S_NTSTRIDE}, s{S_NC}, 0xc000")
    E(f"s_lshl_b32 s{S_N64X4}, s{S_NBLOCK}, {8 if NTB() == 2 else 7}")     # byte offset of the workgroup's first output channel
    E(f"s_lshl_b32 s{S_LD4}, s{S_LDOUT}, 2")
    E(f"s_mul_i32 s{S_SW4}, s{S_W}, s{S_LD4}")
    E(f"s_add_u32 s{S_SUMOFF}, s{S_SW4}, s{S_LD4}")
    # descriptors: input (base per patch), output (base per patch), pooled, scale, shift, weight pieces
    E(f"s_mov_b32 s{S_INR + 2}, s{S_IMGB}")
    E(f"s_mov_b32 s{S_INR + 3}, 0x00020000")
    E(f"s_mov_b32 s{S_OUTR + 2}, s{S_OUTIMGB}")
    E(f"s_mov_b32 s{S_OUTR + 3}, 0x00020000")
    E(f"s_mov_b32 s{S_POOLR + 3}, 0x00020000")
    for (r, p) in ((S_SCR, S_SCALE), (S_SHR, S_SHIFT)):
        E(f"s_mov_b32 s{r}, s{p}")
        E(f"s_and_b32 s{r + 1}, s{p + 1}, 0xffff")
        E(f"s_mov_b32 s{r + 2}, 0x7ffffff0")
        E(f"s_mov_b32 s{r + 3}, 0x00020000")
    # wave roles
    E(f"v_lshrrev_b32_e32 v0, 6, v{VTID}")
    E("s_nop 3")                                    # VALU write -> v_readfirstlane of the same register: wait states (measured: without
    #                                                 them some waves read the OLD v0 = the thread id)
    E("v_readfirstlane_b32 s60, v0")
    E("s_nop 3")
    E(f"s_and_b32 s{S_WI}, s60, 3")
    E(f"s_lshr_b32 s{S_JP}, s60, 2")
    E(f"s_lshl_b32 s61, s{S_WI}, 1")
    E("s_lshr_b32 s62, 0x64, s61")
    E("s_and_b32 s62, s62, 3")                      # ra
    E("s_lshr_b32 s63, 0xda, s61")
    E("s_and_b32 s63, s63, 3")                      # rb
    E(f"s_cmp_eq_u32 s{S_WI}, 1")
    E(f"s_cselect_b32 s{S_SGN}, 1.0, -1.0")
    # weight descriptor: base = wu + nblock * 2 * nC * 49152 + (wi * 4 + 2 jp) * 3072
    E(f"s_lshl_b32 s64, s{S_NTSTRIDE}, {1 if NTB() == 2 else 0}")   # u_bytes of the workgroup's n tiles
    E(f"s_mul_i32 s65, s{S_NBLOCK}, s64")
    E(f"s_lshl_b32 s66, s{S_WI}, 2")
    E(f"s_lshl_b32 s67, s{S_JP}, 1")
    E("s_add_u32 s66, s66, s67")
    E("s_mul_i32 s66, s66, 0xc00")
    E("s_add_u32 s65, s65, s66")
    E(f"s_add_u32 s{S_UR}, s{S_WU}, s65")
    E(f"s_addc_u32 s{S_UR + 1}, s{S_WU + 1}, 0")
    E(f"s_and_b32 s{S_UR + 1}, s{S_UR + 1}, 0xffff")
    E(f"s_mov_b32 s{S_UR + 2}, s64")
    E(f"s_mov_b32 s{S_UR + 3}, 0x00020000")
    # lane constants
    E(f"v_and_b32_e32 v1, 63, v{VTID}")             # lane
    E(f"v_lshlrev_b32_e32 v{VLANE16}, 4, v1")
    E("v_and_b32_e32 v2, 31, v1")                   # lr
    E("v_lshrrev_b32_e32 v3, 5, v1")                # lh
    E("v_and_b32_e32 v4, 15, v2")                   # tx
    E("v_lshrrev_b32_e32 v5, 4, v2")                # ty
    E("v_lshlrev_b32_e32 v5, 1, v5")                # 2 ty
    E("v_mul_u32_u24_e32 v4, 0x50, v4")             # tx * 80
    E("v_lshl_add_u32 v4, v3, 5, v4")               # + lh * 32
    for (dst, srow) in ((VA, 62), (VB, 63)):
        E(f"v_add_u32_e32 v6, s{srow}, v5")
        E("v_mul_u32_u24_e32 v6, 0xaa0, v6")        # (2 ty + r) * 34 * 80
        E("v_add_u32_e32 v6, v6, v4")
        E(f"v_add_u32_e32 v{dst}, 0x{BUFX:x}, v6")  # chunk 0 of a patch sits in slot 4
    for i in range(3):
        E(f"v_lshrrev_b32_e32 v1, 2, v{VTID}")
        if i:
            E(f"v_add_u32_e32 v1, {128 * i}, v1")
        E("v_mul_u32_u24_e32 v2, 0x788, v1")
        E("v_lshrrev_b32_e32 v2, 16, v2")           # r
        E("v_mul_u32_u24_e32 v3, 34, v2")
        E("v_sub_u32_e32 v3, v1, v3")               # cc
        E("v_and_b32_e32 v4, 1, v3")
        E("v_lshl_add_u32 v4, v2, 1, v4")           # r * 2 + (cc & 1)
        E("v_mul_u32_u24_e32 v4, 17, v4")
        E("v_lshrrev_b32_e32 v3, 1, v3")
        E("v_add_u32_e32 v4, v4, v3")
        E("v_mul_u32_u24_e32 v4, 0x50, v4")
        E(f"v_and_b32_e32 v5, 3, v{VTID}")
        E(f"v_lshl_add_u32 v{VHST[i]}, v5, 4, v4")
    E(f"v_mov_b32_e32 v{VMASK()}, s{S_MASK}")
    E(f"v_mov_b32_e32 v{VSGN()}, s{S_SGN}")
    # pipeline lead-in
    E(f"s_mov_b32 s{S_LC}, 0")
    E(f"s_mov_b32 s{S_LP}, 0")
    E(f"s_mov_b32 s{S_PI}, 0")
    E(f"s_mov_b32 s{S_LC64}, 0")
    if NTB() == 1:
        # the wave's weight pieces of every chunk (resident for the life of the workgroup), then the halo of chunks 0 (-> slot 4), 1 and 2
        for c in range(CFG["nc"]):
            for jj in range(2):
                E(f"s_mov_b32 s{S_WO[0][0]}, 0x{c * 0xc000 + jj * 0xc00:x}")
                for pp in range(3):
                    E(f"buffer_load_dwordx4 {vr(WN(c, jj, pp), 4)}, v{VLANE16}, s[{S_UR}:{S_UR + 3}], s{S_WO[0][0]} offen offset:{pp * 1024}")
        setup_load()
        halo_loads(True, 0)
        for i in range(3):
            E(f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}")
        E("s_waitcnt vmcnt(0)")
        halo_stores(0)
        for i in range(3):
            E(f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}")
        for setn in (1, 0):
            lsk = newlabel("nosetup")
            E(f"s_cmp_lg_u32 s{S_LC}, 0")
            E(f"s_cbranch_scc1 {lsk}")
            E(f"s_cmp_ge_u32 s{S_LP}, s{S_NPATCH}")
            E(f"s_cbranch_scc1 {lsk}")
            setup_load()
            L(lsk)
            halo_loads(True, setn)
        E("s_waitcnt lgkmcnt(0)")
        E("s_barrier")
        return
    setup_load()
    halo_loads()                                     # chunk 0
    for i in range(3):
        E(f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}")
    E("s_waitcnt vmcnt(0)")
    halo_stores()                                    # -> slot 4
    for i in range(3):
        E(f"v_xor_b32_e32 v{VHST[i]}, 0x{BUFX:x}, v{VHST[i]}")
    lskip = newlabel("nosetup")
    E(f"s_cmp_lg_u32 s{S_LC}, 0")
    E(f"s_cbranch_scc1 {lskip}")
    E(f"s_cmp_ge_u32 s{S_LP}, s{S_NPATCH}")
    E(f"s_cbranch_scc1 {lskip}")
    setup_load()
    L(lskip)
    halo_loads()                                     # chunk 1 -> halo registers
    # weight pieces of chunk 0
    E(f"s_mov_b32 s{S_WO[0][0]}, 0")
    E(f"s_mov_b32 s{S_WO[0][1]}, s{S_NTSTRIDE}")
    E(f"s_mov_b32 s{S_WO[1][0]}, 0xc00")
    E(f"s_add_u32 s{S_WO[1][1]}, s{S_NTSTRIDE}, 0xc00")
    for jj in range(2):
        for nt in range(2):
            for p in range(3):
                E(weight_load(jj, nt, p))
    if opt("zero_acc") or opt("no_epilogue"):
        for r in range(128):
            E(f"v_mov_b32_e32 v{r}, 0")
    E("s_waitcnt lgkmcnt(0)")
    E("s_barrier")


def emit_dump(first_reg):
    """debug: thread tid stores 32 consecutive registers at out + tid * 128 bytes and the program ends"""
    E("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E("s_nop 7")
    E("s_nop 7")
    E(f"s_mov_b32 s{S_OUTR}, s{S_OUT}")
    E(f"s_and_b32 s{S_OUTR + 1}, s{S_OUT + 1}, 0xffff")
    E(f"s_mov_b32 s{S_OUTR + 2}, 0x7ffffff0")
    E(f"s_mov_b32 s{S_OUTR + 3}, 0x00020000")
    E(f"v_lshlrev_b32_e32 v{TT(0)}, 7, v{VTID}")
    for k in range(8):
        E(f"buffer_store_dwordx4 {vr(first_reg + 4 * k, 4)}, v{TT(0)}, s[{S_OUTR}:{S_OUTR + 3}], 0 offen offset:{16 * k}")
    E("s_waitcnt vmcnt(0)")
    E("s_endpgm")


def emit_first_form(jp):
    """step 0 of the first chunk (no MFMAs to hide behind)"""
    if opt("pingpong") and jp == 1:       # waves 4-7 transform inside the loop: only the request
        for r in raw_reads(jp, 0, 0):
            E(r)
        return
    for r in raw_reads(jp, 0, 0):
        E(r)
    E("s_waitcnt lgkmcnt(0)")
    for hf in range(2):
        for x in form_valu(jp, 0, 0, hf):
            E(x)
    if opt("spread", 1) and not opt("pingpong"):
        for r in raw_reads(jp, 0, 1)[0:4]:
            E(r)


def emit_kernel(name):
    _lbl[0] += 1000
    end = f".Lend_{name}"
    jp1 = f".Ljp1_{name}"
    fe = f".Lfunc_end_{name}"
    hdr = f"""\t.text
\t.protected\t{name}
\t.globl\t{name}
\t.p2align\t8
\t.type\t{name},@function
{name}:"""
    out.append(hdr)
    global END_LABEL
    END_LABEL = end
    emit_prologue()
    if DEBUG.startswith("dump_pro:"):      # registers after the lead-in (raw reads of step 0 are issued but not transformed)
        for r in raw_reads(0, 0, 0):
            E(r)
        for i, sr in enumerate((S_SGN, S_WI, S_JP, 60, 61, 62, 63, S_NBLOCK, S_PBEGIN, S_NPATCH, S_LC, S_LP, S_NC, S_UR, S_UR + 1, S_UR + 2)):
            E(f"v_mov_b32_e32 v{224 + i}, s{sr}")
        emit_dump(int(DEBUG.split(":")[1]))
    E(f"s_cmp_lg_u32 s{S_JP}, 0")
    E(f"s_cbranch_scc1 {jp1}")
    if opt("prio_jp0"):
        E(f"s_setprio {opt('prio_jp0')}")
    emit_first_form(0)
    (emit_patch_loop if NTB() == 2 else emit_patch_loop_n)(0)
    L(jp1)
    if opt("prio_jp1"):      # static priority for the second-dispatched half (the SIMD partners of waves 0-3)
        E(f"s_setprio {opt('prio_jp1')}")
    emit_first_form(1)
    (emit_patch_loop if NTB() == 2 else emit_patch_loop_n)(1)
    L(end)
    E("s_endpgm")
    out.append(f"""\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size 163840
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size 120
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_dispatch_ptr 0
\t\t.amdhsa_user_sgpr_queue_ptr 0
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_user_sgpr_dispatch_id 0
\t\t.amdhsa_user_sgpr_kernarg_preload_length 0
\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0
\t\t.amdhsa_user_sgpr_private_segment_size 0
\t\t.amdhsa_uses_dynamic_stack 0
\t\t.amdhsa_enable_private_segment 0
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 0
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_sgpr_workgroup_info 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 256
\t\t.amdhsa_next_free_sgpr 102
\t\t.amdhsa_accum_offset 256
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t\t.amdhsa_fp16_overflow 0
\t\t.amdhsa_tg_split 0
\t.end_amdhsa_kernel
\t.text
{fe}:
\t.size\t{name}, {fe}-{name}
""")
    META.append(f"""  - .agpr_count:     0
    .args:
      - .offset:         0
        .size:           120
        .value_kind:     by_value
    .group_segment_fixed_size: 163840
    .kernarg_segment_align: 8
    .kernarg_segment_size: 120
    .max_flat_workgroup_size: 512
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     108
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     256
    .vgpr_spill_count: 0
    .wavefront_size: 64""")


META = []
END_LABEL = ".Lend"
VARIANTS = [   # (kernel-name suffix, options); suffix "" = the shipping kernel
    ("", {}),
]
if os.environ.get("GEN_WINO_VARIANTS"):
    VARIANTS += [
        ("_v1", {"no_epilogue": 1}),
        ("_v2", {"no_epilogue": 1, "no_valu": 1}),
        ("_v3", {"no_epilogue": 1, "no_mfma": 1}),
        ("_v4", {"no_epilogue": 1, "no_barrier": 1}),
        ("_v5", {"slow_valu": 1}),
        ("_v6", {"sp_v0": (1, 5), "sp_v1": (6, 10)}),
        ("_v7", {"prio": ((0, 0, 1, 1), (1, 1, 0, 0))}),
        ("_v8", {"no_epilogue": 1, "stamp": 1}),
        ("_v9", {"no_epilogue": 1, "no_valu": 1, "stamp": 1}),
        ("_v10", {"no_epilogue": 1, "no_mfma": 1, "stamp": 1}),
        ("_v11", {"no_epilogue": 1, "no_wload": 1, "stamp": 1}),
        ("_v12", {"no_epilogue": 1, "no_halo": 1, "stamp": 1}),
        ("_v13", {"no_epilogue": 1, "no_ldsread": 1, "stamp": 1}),
        ("_v15", {"no_epilogue": 1, "steptimes": 1}),
        ("_v16", {"no_epilogue": 1, "steptimes": 1, "slow_valu": 1}),
        ("_v17", {"prio": ((0, 0, 0, 0), (0, 1, 0, 1))}),
        ("_v18", {"sp_v0": (0, 5), "sp_v1": (6, 11)}),
        ("_v19", {"sp_v0": (0, 5), "sp_v1": (6, 11), "prio": ((0, 0, 0, 0), (1, 0, 1, 0))}),
        ("_v20", {"no_epilogue": 1, "steptimes": 1, "sp_v0": (0, 5), "sp_v1": (6, 11)}),
        ("_v21", {"zero_acc": 1}),
        ("_v14", {"no_epilogue": 1, "no_wload": 1, "no_halo": 1, "no_ldsread": 1, "no_valu": 1, "no_barrier": 1, "stamp": 1}),
    ]


def main():
    path = sys.argv[1]
    out.append('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
    for suffix, o in VARIANTS:
        OPT.clear()
        OPT.update(o)
        CFG.update(ntb=2, nc=None)
        emit_kernel("mgu_wino_cp2_gfx950" + suffix)
    for nc in (2, 4):            # the narrow kernels: 32 output channels, the layer's 2 / 4 chunks of weight pieces resident
        OPT.clear()
        CFG.update(ntb=1, nc=nc)
        emit_kernel(f"mgu_wino_cp1r{nc}_gfx950")
    out.append("\t.amdgpu_metadata\n---\namdhsa.kernels:")
    out.extend(META)
    out.append("""amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...

\t.end_amdgpu_metadata
""")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
