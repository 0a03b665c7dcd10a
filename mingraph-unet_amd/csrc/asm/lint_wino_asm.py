#!/usr/bin/env python3
"""Static check of the hand-counted waits and wait states of the generated Winograd assembly (asm/gen_wino_cp.py).

The hand-written kernels carry no compiler-inserted s_waitcnt / s_nop: every wait is counted by the generator.  This checker
replays the instruction stream of each kernel's steady-state loops (the loop bodies repeated, branches taken as straight-line code) with the machine's
in-order counters and fails on:
  * a read of a VGPR that is still the destination of an outstanding buffer load or ds_read (vmcnt / lgkmcnt retire in order; the
    model waits exactly as the s_waitcnt in the stream says);
  * a write of a VGPR that is the destination of an outstanding load (the load would land on top of the new value);
  * an MFMA that reads a VGPR written by a VALU instruction fewer than 2 wait states earlier;
  * a non-MFMA read of an MFMA result fewer than 18 wait states after the MFMA that wrote it;
  * a VALU write of the data registers of a 16-byte LDS / buffer store in the very next instruction;
  * an LDS add-TID store directly behind the s_mov that wrote M0; v_readfirstlane directly behind the VALU write of its source.
Used by tests/test_asm_lint.py (CPU) and runnable by hand:  lint_wino_asm.py FILE.s
"""
import re
import sys


def regs(tok):
    tok = tok.strip().lstrip("-").strip("|")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    args = []
    if len(parts) > 1:
        a = re.split(r",\s*", parts[1])
        # trailing modifiers ride on the last operand ("s41 offen offset:16 nt")
        args = [x.split()[0] if x.split() else x for x in a]
    return op, args, line


def classify(op, args):
    """-> (dst regs, src regs, kind)"""
    if op.startswith("v_mfma"):
        return regs(args[0]), regs(args[1]) | regs(args[2]) | regs(args[3]), "mfma"
    if op.startswith("ds_read"):
        return regs(args[0]), regs(args[1]), "lds_load"
    if op.startswith("ds_write_addtid"):
        return set(), regs(args[0]), "lds_store_addtid"
    if op.startswith("ds_write"):
        s = set()
        for a in args:
            s |= regs(a)
        return set(), s, "lds_store"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return regs(args[0]), regs(args[1]), "vm_load"
    if op.startswith("buffer_store") or op.startswith("global_store"):
        return set(), regs(args[0]) | regs(args[1]), "vm_store"
    if op.startswith("v_readfirstlane"):
        return set(), regs(args[1]), "readlane"
    if op.startswith("v_cmp"):
        s = set()
        for a in args:
            s |= regs(a)
        return set(), s, "valu"
    if op.startswith("v_"):
        s = set()
        for a in args[1:]:
            s |= regs(a)
        return regs(args[0]), s, "valu"
    return set(), set(), "other"


def kernels(text):
    """name -> list of source lines (one kernel each)"""
    out, cur, name = {}, None, None
    for ln in text.split("\n"):
        m = re.match(r"^(mgu_\w+):\s*$", ln)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None:
            cur.append(ln)
    return out


def loop_bodies(lines):
    """[(label, body lines)] of the backward branches of a kernel: the text between a label and the LAST branch back to it"""
    lab = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.\w+):", ln)
        if m:
            lab[m.group(1)] = i
    bodies = []
    for i, ln in enumerate(lines):
        m = re.match(r"\s*s_cbranch_scc1\s+(\.\w+)", ln)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            bodies.append((m.group(1), lines[lab[m.group(1)]:i]))
    return bodies


def replay(stream, errors, where, prime=()):
    vm, lg = [], []                 # outstanding (dst regs, text) in issue order
    valu_age = {}                   # reg -> wait states since a VALU wrote it
    mfma_age = {}                   # reg -> wait states since an MFMA wrote it
    prev = None
    m0_fresh = False

    def age(n):
        for d in (valu_age, mfma_age):
            for r in list(d):
                d[r] += n
                if d[r] > 40:
                    del d[r]

    skip_to = None
    labels = {}
    for i, ln in enumerate(stream):
        m = re.match(r"^(\.\w+):", ln)
        if m:
            labels.setdefault(m.group(1), []).append(i)
    for i, ln in enumerate(stream):
        if skip_to is not None:          # behind an unconditional forward branch: resume at its label
            if re.match(r"^" + re.escape(skip_to) + ":", ln):
                skip_to = None
            continue
        p = parse(ln)
        if p is None:
            continue
        op, args, text = p
        if op == "s_branch" and any(j > i for j in labels.get(args[0], [])):
            skip_to = args[0]
            continue
        if op == "s_nop":
            age(int(args[0]) + 1)
            prev = ("nop", set(), text)
            m0_fresh = False
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", text)
            if m:
                del vm[:max(0, len(vm) - int(m.group(1)))]
            m = re.search(r"lgkmcnt\((\d+)\)", text)
            if m:
                del lg[:max(0, len(lg) - int(m.group(1)))]
            age(1)
            prev = ("wait", set(), text)
            continue
        dst, src, kind = classify(op, args)
        touched = dst | src
        for q, nm in ((vm, "buffer load"), (lg, "LDS read")):
            for (d, t) in q:
                if d & src:
                    errors.append(f"{where}: `{text}` reads v{sorted(d & src)[0]} while `{t}` ({nm}) is outstanding")
                if d & dst and kind not in ("vm_load", "lds_load"):
                    errors.append(f"{where}: `{text}` writes v{sorted(d & dst)[0]}, destination of outstanding `{t}`")
        if kind == "mfma":
            for r in src:
                if valu_age.get(r, 99) < 2:
                    errors.append(f"{where}: `{text}` reads v{r} {valu_age[r]} wait state(s) after a VALU write (2 required)")
        else:
            for r in touched if kind in ("valu", "lds_store", "lds_store_addtid", "vm_store", "readlane") else src:
                if r in src and mfma_age.get(r, 99) < 18:
                    errors.append(f"{where}: `{text}` reads MFMA result v{r} after {mfma_age[r]} wait states (18 required)")
        if kind == "valu" and prev and prev[0] in ("lds_store", "vm_store") and len(prev[1]) >= 4 and (dst & prev[1]):
            errors.append(f"{where}: `{text}` overwrites store data of the previous instruction `{prev[2]}`")
        if kind == "lds_store_addtid" and m0_fresh:
            errors.append(f"{where}: `{text}` directly behind the write of M0")
        if kind == "readlane" and any(valu_age.get(r, 99) < 1 for r in src):
            errors.append(f"{where}: `{text}` directly behind the VALU write of its source")
        age(1)
        if kind == "valu":
            for r in dst:
                valu_age[r] = 0
                mfma_age.pop(r, None)
        if kind == "mfma":
            for r in dst:
                mfma_age[r] = 0
                valu_age.pop(r, None)
        if kind == "vm_load":
            vm.append((dst, text))
        elif kind == "vm_store":
            vm.append((set(), text))
        elif kind == "lds_load":
            lg.append((dst, text))
        elif kind in ("lds_store", "lds_store_addtid"):
            lg.append((set(), text))
        m0_fresh = op == "s_mov_b32" and args and args[0] == "m0"
        data = set()
        if kind in ("lds_store", "vm_store"):
            data = regs(args[1]) if kind == "lds_store" else regs(args[0])
        prev = (kind, data, text)


def check(text):
    errors = []
    ks = kernels(text)
    for name, lines in ks.items():
        # whole kernel once (prologue, one trip through everything), then every loop body three times in a row
        replay(lines, errors, f"{name} (straight line)")
        for lab, body in loop_bodies(lines):
            replay(body * 3, errors, f"{name} loop {lab}")
    return errors, len(ks)


if __name__ == "__main__":
    errs, n = check(open(sys.argv[1]).read())
    for e in errs[:40]:
        print(e)
    print(f"{n} kernels, {len(errs)} finding(s)")
    sys.exit(1 if errs else 0)
