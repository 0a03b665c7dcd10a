"""The hand-scheduled Winograd assembly (csrc/asm/gen_wino_cp.py) carries no compiler-inserted waits: every s_waitcnt count and
every wait state is the generator's.  csrc/asm/lint_wino_asm.py replays the generated stream with the machine's in-order counters;
this test (CPU only: it needs neither a GPU nor the assembler) runs it on what the generator emits now, on the timing-only variants,
and on deliberately broken streams -- the lint must pass the first two and catch the seeded faults."""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "mingraph-unet_amd", "csrc", "asm")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ASM, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _generate(tmp_path, variants=False):
    out = tmp_path / ("v.s" if variants else "k.s")
    env = dict(os.environ)
    env.pop("GEN_WINO_DEBUG", None)
    if variants:
        env["GEN_WINO_VARIANTS"] = "1"
    else:
        env.pop("GEN_WINO_VARIANTS", None)
    subprocess.run([sys.executable, os.path.join(ASM, "gen_wino_cp.py"), str(out)], check=True, env=env)
    return out.read_text()


def test_generated_streams_pass_the_lint(tmp_path):
    lint = _load("lint_wino_asm")
    text = _generate(tmp_path)
    errs, n = lint.check(text)
    assert n == 3 and not errs, errs[:5]          # the wide kernel and the two narrow kernels
    # two component-pair code paths each; wide: the peeled first chunk + the chunk loop (48 MFMAs per chunk); narrow: 24 per unrolled chunk
    assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", text)) == 2 * (2 * 48 + 24 * 2 + 24 * 4)
    # every accumulator chain of a patch starts from the constant 0 (no clearing pass): 8 tuples in the wide kernel, 4 in a narrow one
    assert len(re.findall(r"v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\], 0\n", text)) == 2 * (8 + 4 + 4)


def test_shipping_kernels_of_a_variants_build_pass_the_lint(tmp_path):
    lint = _load("lint_wino_asm")
    text = _generate(tmp_path, variants=True)
    ks = lint.kernels(text)
    ship = {k: v for k, v in ks.items() if not re.search(r"_v\d+$", k)}
    assert len(ship) == 3 and len(ks) > 10
    errs = []
    for name, lines in ship.items():
        lint.replay(lines, errs, name)
        for lab, body in lint.loop_bodies(lines):
            lint.replay(body * 3, errs, f"{name} {lab}")
    assert not errs, errs[:5]


def test_lint_catches_seeded_faults(tmp_path):
    lint = _load("lint_wino_asm")
    text = _generate(tmp_path)
    weak_wait = text.replace("s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(9)")          # the wide kernel's step-0 weight wait, as it once was
    assert weak_wait != text and lint.check(weak_wait)[0]
    no_states = "\n".join(l for l in text.split("\n") if l.strip() != "s_nop 1")   # VALU -> MFMA wait states, store-data distance
    assert lint.check(no_states)[0]
    weak_lds = text.replace("s_waitcnt lgkmcnt(0)", "s_waitcnt lgkmcnt(2)")
    assert lint.check(weak_lds)[0]
