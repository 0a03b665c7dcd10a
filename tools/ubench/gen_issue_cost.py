#!/usr/bin/env python3
"""Micro-benchmark generator: issue cost of the VALU instructions of the Winograd transform / split, alone and in the gaps of a dependent
v_mfma_f32_32x32x16_bf16 chain (one wave per SIMD and two waves per SIMD).  Emits one kernel per case: `ub_<case>`; each runs REP repeats of
its body between two s_memtime stamps and stores the cycle count per wave to out[wave_global].
usage: gen_issue_cost.py OUT.s ; build: clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c OUT.s -o o.co && ld.lld -shared o.co -o OUT.hsaco"""
import sys
REP = 200
CASES = {}
def valu_block(kind, n, base=64):
    r = []
    for i in range(n):
        d = base + (i % 16)
        if kind == "fma_s":
            r.append(f"v_fma_f32 v{d}, s20, v{d + 16}, v{d}")
        elif kind == "fma_v":
            r.append(f"v_fma_f32 v{d}, v{d + 32}, v{d + 16}, v{d}")
        elif kind == "perm":
            r.append(f"v_perm_b32 v{d}, v{d + 16}, v{d + 32}, s21")
        elif kind == "and":
            r.append(f"v_and_b32_e32 v{d}, s22, v{d + 16}")
        elif kind == "sub":
            r.append(f"v_sub_f32_e32 v{d}, v{d + 16}, v{d + 32}")
        elif kind == "add":
            r.append(f"v_add_f32_e32 v{d}, v{d + 16}, v{d + 32}")
        elif kind == "pack":
            r.append(f"v_pack_b32_f16 v{d}, v{d + 16}, v{d + 32} op_sel:[1,1,0]")
        elif kind == "cvtpk":
            r.append(f"v_cvt_pk_bf16_f32 v{d}, v{d + 16}, v{d + 32}")
        elif kind == "fmac":      # VOP2, VGPR operands only
            r.append(f"v_fmac_f32_e32 v{d}, v{d + 32}, v{d + 16}")
        elif kind == "andv":      # VOP2 and with the mask in a VGPR
            r.append(f"v_and_b32_e32 v{d}, v{d + 32}, v{d + 16}")
        elif kind == "pkadd":     # VOP3P packed fp32: two adds per lane and instruction on aligned register pairs
            dd = base + 2 * (i % 8)
            r.append(f"v_pk_add_f32 v[{dd}:{dd + 1}], v[{dd + 16}:{dd + 17}], v[{dd + 32}:{dd + 33}] neg_lo:[0,1] neg_hi:[0,1]")
        elif kind == "pkfma":     # packed fma, one factor broadcast from the low half of a pair
            dd = base + 2 * (i % 8)
            r.append(f"v_pk_fma_f32 v[{dd}:{dd + 1}], v[{dd + 16}:{dd + 17}], v[{dd + 32}:{dd + 33}], v[{dd}:{dd + 1}] op_sel_hi:[1,0,1]")
        elif kind == "dot2c":     # VOP2 v_dot2c_f32_bf16: d += a.bf16x2 . b.bf16x2 (the residual v - hi straight from the packed piece)
            r.append(f"v_dot2c_f32_bf16_e32 v{d}, v{d + 32}, v{d + 16}")
        elif kind == "dot2cs":    # the constant pair in an SGPR
            r.append(f"v_dot2c_f32_bf16_e32 v{d}, s23, v{d + 16}")
        elif kind == "mix4":      # transform + split with the residuals by v_dot2c: 8 fmac + 4 add, 6 perm, 8 dot2c = 26 per half
            seq = ["fmac"] * 8 + ["add"] * 4 + ["perm"] * 2 + ["dot2c"] * 4 + ["perm"] * 2 + ["dot2c"] * 4 + ["perm"] * 2
            return [valu_block(seq[i % 26], 1, base + (i % 12))[0] for i in range(n)]
        elif kind == "mix3":      # the transform + split on packed adds: per 8 values 4 pk transform, 8 and, 4 pk sub, 8 and, 4 pk sub, 12 perm -> 26 per 8
            seq = ["pkfma"] * 4 + ["pkadd"] * 2 + ["perm"] * 2 + ["andv"] * 4 + ["pkadd"] * 2 + ["perm"] * 2 + ["andv"] * 4 + ["pkadd"] * 2 + ["perm"] * 2
            return [valu_block(seq[i % 24], 1, base + 2 * (i % 6))[0] for i in range(n)]
        elif kind == "mix2":      # the transform + split with VOP2 / VGPR-only forms: 8 fmac + 4 add, 6 perm, 8 and (VGPR mask), 8 sub
            seq = ["fmac"] * 8 + ["add"] * 4 + ["perm"] * 2 + ["andv"] * 4 + ["sub"] * 4 + ["perm"] * 2 + ["andv"] * 4 + ["sub"] * 4 + ["perm"] * 2
            return [valu_block(seq[i % 34], 1, base + (i % 12))[0] for i in range(n)]
        elif kind == "mix":      # the transform + split mix of one half: 12 fma, 6 perm, 8 and, 8 sub
            seq = ["fma_s"] * 12 + ["perm"] * 2 + ["and"] * 4 + ["sub"] * 4 + ["perm"] * 2 + ["and"] * 4 + ["sub"] * 4 + ["perm"] * 2
            return [valu_block(seq[i % 34], 1, base + (i % 12))[0] for i in range(n)]
    return r
MF = "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]"
MF2 = "v_mfma_f32_32x32x16_bf16 v[32:47], v[16:19], v[20:23], v[32:47]"
for k in ("fma_s", "perm", "and", "sub", "fmac", "andv", "pkadd", "pkfma", "dot2c", "dot2cs", "mix", "mix2", "mix3", "mix4"):
    CASES[f"valu_{k}"] = valu_block(k, 48)                         # 48 VALU alone
    body = []
    for i in range(8):
        body.append(MF)
        body += valu_block(k, 6, 64 + 0)                            # 8 MFMAs, 6 VALU per gap
    CASES[f"mfma6_{k}"] = body
    body = []
    for i in range(8):
        body.append(MF)
        body += valu_block(k, 8, 64)
    CASES[f"mfma8_{k}"] = body
# accumulators in AccVGPRs (accum_offset 128: v0-127 + a0-127): does the VALU in the shadow of an MFMA get cheaper when C / D are not in the
# architectural file?
MFA = "v_mfma_f32_32x32x16_bf16 a[0:15], v[16:19], v[20:23], a[0:15]"
ACC_CASES = set()
for k in ("fmac", "mix2"):
    for g in (6, 8):
        body = []
        for i in range(8):
            body.append(MFA)
            body += valu_block(k, g, 64)
        CASES[f"mfma{g}acc_{k}"] = body
        ACC_CASES.add(f"mfma{g}acc_{k}")
CASES["mfma_only_acc"] = [MFA] * 8
ACC_CASES.add("mfma_only_acc")
CASES["mfma_only"] = [MF] * 8
CASES["mfma_2acc"] = [MF, MF2] * 4
body = []
for i in range(8):                                                  # burst of 8 MFMAs then 48 VALU (phase form)
    body.append(MF)
body += valu_block("mix", 48)
CASES["burst_then_mix"] = body

out = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"']
meta = []
for name, body in CASES.items():
    k = "ub_" + name
    out.append(f"\t.text\n\t.protected {k}\n\t.globl {k}\n\t.p2align 8\n\t.type {k},@function\n{k}:")
    out += ["\ts_load_dwordx2 s[4:5], s[0:1], 0x0", "\ts_mov_b32 s20, 1.0", "\ts_mov_b32 s21, 0x07060302", "\ts_mov_b32 s22, 0xffff0000", "\ts_mov_b32 s23, 0x0000bf80",
            "\tv_mov_b32 v200, v0"]
    for r in range(24, 128):
        out.append(f"\tv_mov_b32_e32 v{r}, 1.0")
    if name in ACC_CASES:
        for r in range(0, 16):
            out.append(f"\tv_accvgpr_write_b32 a{r}, 0")
    for r in range(0, 24):
        out.append(f"\tv_mov_b32_e32 v{r}, 0")
    out += ["\ts_waitcnt lgkmcnt(0)", "\ts_barrier", "\ts_memtime s[8:9]", "\ts_waitcnt lgkmcnt(0)", f"\ts_movk_i32 s10, {REP}", f".L{k}:"]
    out += ["\t" + x for x in body]
    out += ["\ts_sub_u32 s10, s10, 1", "\ts_cmp_lg_u32 s10, 0", f"\ts_cbranch_scc1 .L{k}", "\ts_nop 7", "\ts_nop 7", "\ts_memtime s[12:13]",
            "\ts_waitcnt lgkmcnt(0)", "\ts_sub_u32 s12, s12, s8",
            "\tv_lshrrev_b32_e32 v201, 6, v200", "\ts_lshl_b32 s14, s2, 4", "\tv_add_u32_e32 v201, s14, v201", "\tv_lshlrev_b32_e32 v201, 2, v201",
            "\tv_mov_b32_e32 v202, s12", "\tv_and_b32_e32 v203, 63, v200", "\tv_cmp_eq_u32_e32 vcc, 0, v203", "\ts_and_b64 exec, exec, vcc",
            "\tglobal_store_dword v201, v202, s[4:5]", "\ts_waitcnt vmcnt(0)", "\ts_endpgm"]
    out.append(f"""\t.section .rodata,"a",@progbits
\t.p2align 6, 0x0
\t.amdhsa_kernel {k}
\t\t.amdhsa_group_segment_fixed_size 0
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size 8
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 256
\t\t.amdhsa_next_free_sgpr 32
\t\t.amdhsa_accum_offset {128 if name in ACC_CASES else 256}
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.text""")
    meta.append(f"""  - .agpr_count: {128 if name in ACC_CASES else 0}
    .args:
      - .offset: 0
        .size: 8
        .value_kind: global_buffer
        .address_space: global
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 8
    .max_flat_workgroup_size: 1024
    .name: {k}
    .private_segment_fixed_size: 0
    .sgpr_count: 40
    .sgpr_spill_count: 0
    .symbol: {k}.kd
    .vgpr_count: {128 if name in ACC_CASES else 256}
    .vgpr_spill_count: 0
    .wavefront_size: 64""")
out.append("\t.amdgpu_metadata\n---\namdhsa.kernels:")
out += meta
out.append("amdhsa.target: amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\t.end_amdgpu_metadata")
open(sys.argv[1], "w").write("\n".join(out) + "\n")
open(sys.argv[1] + ".cases", "w").write("\n".join(f"{n} {sum(1 for x in b if 'mfma' in x)} {sum(1 for x in b if 'mfma' not in x)}" for n, b in CASES.items()) + "\n")
