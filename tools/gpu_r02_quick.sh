#!/usr/bin/env bash
# Quick A/B session: kernel + parity tests, then the headline bench under the given environment variants.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r02q}
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_backward_kernels.py tests/test_gpu_train.py -q -m gpu -p no:cacheprovider --maxfail=20 ${PYTEST_ARGS:-} > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; tail -n 15 gpurun_out/${TAG}_tests.log
i=0
for env in "${@:-X=1}"; do
  i=$((i+1))
  env $env timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench_$i.log 2>&1
  echo "== bench [$env] rc=$?"
  python - <<PY
import json
try:
    j=json.loads([l for l in open("gpurun_out/${TAG}_bench_$i.log") if l.startswith("{")][-1])
    print(j["value"], "Mpix/s", j["ms_per_step"], "ms", j.get("spread"))
    for k in (j.get("roofline") or {}).get("kernels", []): print("   ", k["kernel"], k["launches_per_step"], k["ms_per_step"], k["avg_launch_us"], k["issued_tflops"])
    g=j.get("gat") or {}
    for key in ("graphs_8","graphs_64"):
        if key in g:
            for sch in ("aggregate_first","wh_row_gather"): print("   gat", key, sch, g[key][sch]["layer_us"], g[key][sch]["kernel_us"], g[key][sch]["frac_of_hbm_peak"])
    if "error" in g: print("gat error", g["error"])
except Exception as e:
    print("no json", e); print(open("gpurun_out/${TAG}_bench_$i.log").read()[-1500:])
PY
done
