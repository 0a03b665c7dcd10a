"""Runs the issue-cost micro-benchmarks (tools/ubench/gen_issue_cost.py): one workgroup per CU of 256 threads (one wave per SIMD) and of 512
threads (two waves per SIMD); prints cycles per loop body and per instruction.
Build (in the build container; the .s and the .hsaco are git-ignored, the .hsaco travels with the gpurun snapshot):
  python3 tools/ubench/gen_issue_cost.py tools/ubench/issue_cost.s
  /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c tools/ubench/issue_cost.s -o /tmp/ic.co
  /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/ic.co -o tools/ubench/issue_cost.hsaco"""
import ctypes as C, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
blob = open(os.path.join(here, "issue_cost.hsaco"), "rb").read()
mod = C.c_void_p()
torch.zeros(1, device="cuda")
assert hip.hipModuleLoadData(C.byref(mod), blob) == 0
REP = 200
out = torch.zeros(256 * 16, dtype=torch.int32, device="cuda")
for line in open(os.path.join(here, "issue_cost.s.cases")):
    name, nm, nv = line.split()
    nm, nv = int(nm), int(nv)
    fn = C.c_void_p()
    assert hip.hipModuleGetFunction(C.byref(fn), mod, ("ub_" + name).encode()) == 0
    res = []
    for threads in (256, 512):
        out.zero_()
        ptr = C.c_void_p(out.data_ptr())
        args = (C.c_void_p * 1)(C.cast(C.pointer(ptr), C.c_void_p))
        for _ in range(3):
            assert hip.hipModuleLaunchKernel(fn, 256, 1, 1, threads, 1, 1, 0, None, args, None) == 0
        torch.cuda.synchronize()
        v = out.view(256, 16)[:, :threads // 64].float() / REP
        if threads == 256:
            res.append(f"{float(v.median()):7.1f}")
        else:   # waves 0-3 (older) and 4-7 (their SIMD partners) separately; the slower one is the pair's time
            res.append(f"old {float(v[:, :4].median()):7.1f} young {float(v[:, 4:].median()):7.1f}")
    print(f"{name:18s} mfma {nm:2d} valu {nv:3d}   1 wave/SIMD: {res[0]} cyc/body   2 waves/SIMD: {res[1]}")
