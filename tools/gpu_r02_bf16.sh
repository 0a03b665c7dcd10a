#!/usr/bin/env bash
# bf16 session: bf16 parity tests, then the bf16 bench under the given environment variants.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r02bf}
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu -p no:cacheprovider -s > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; tail -n 12 gpurun_out/${TAG}_tests.log
i=0
for env in "${@:-X=1}"; do
  i=$((i+1))
  env $env timeout -k 10 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench_$i.log 2>&1
  echo "== bench bf16 [$env] rc=$?"
  python - <<PY
import json
try:
    j=json.loads([l for l in open("gpurun_out/${TAG}_bench_$i.log") if l.startswith("{")][-1])
    print(j["value"], "Mpix/s", j["ms_per_step"], "ms", j.get("spread"))
    r=j.get("roofline") or {}
    print("  dominant", r.get("kernel"), r.get("frac"), "all conv alg TF", (r.get("all_conv_kernels") or {}).get("algorithmic_tflops"))
    for k in r.get("kernels", []): print("   ", k["kernel"], k["launches_per_step"], k["ms_per_step"], k["avg_launch_us"], k["issued_tflops"])
except Exception as e:
    print("no json", e); print(open("gpurun_out/${TAG}_bench_$i.log").read()[-1500:])
PY
done
