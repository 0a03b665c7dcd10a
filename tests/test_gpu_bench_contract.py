"""The driver's contract with bench.py: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the agreed keys, the
roofline object of the dominant kernel and (N = 1) the CPU baseline; the train mode does the same for BASELINE configs[4]."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "cpu_baseline"}


def run_bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_forward_line(cuda):
    j = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sustained-seconds", "0")
    assert KEYS <= set(j), sorted(KEYS - set(j))
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["unit"] == "Mpix/s" and j["dtype"] == "f32" and j["vs_baseline"] is None and "workload" in j["config"]
    assert abs(j["value"] - 8 * 512 * 512 / (j["ms_per_step"] * 1e-3) / 1e6) <= 1e-2 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 < r["algorithmic_frac"] < r["frac"]          # Winograd + three-piece split: issued > algorithmic
    assert "3 bf16 pieces" in r["arithmetic"] and "bf16" in r["pipe"]


def test_train_line(cuda):
    j = run_bench("--gpus", "1", "--mode", "train", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert KEYS <= set(j)
    assert j["unit"] == "Mpix/s" and j["steps"] == 2 and "configs[4]" in j["config"]["workload"]
    assert j["roofline"]["kernel"].startswith("wino") and 0.0 < j["roofline"]["frac"] < 1.0
    assert 0.0 < j["final_loss"] < 10.0


def test_plain_run_carries_cpu_baseline_and_other_configs(cuda):
    """What the driver runs at round end (no extra flags): the headline line plus the CPU baseline and the bf16 / train-step side measurements."""
    j = run_bench("--steps", "3", "--warmup", "1", "--sustained-seconds", "0")
    assert KEYS <= set(j)
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == j["unit"] and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    o = j["other_configs"]
    assert not any(k.endswith("_error") for k in o), o
    assert len(o) == 2 and all(v["ms_per_step"] > 0 and v["mpix_per_s"] > 0 for v in o.values())
