#!/usr/bin/env bash
# Round-3 A/B of the TRAIN step: selected tests, then `bench.py --mode train` under each environment variant (alternating A B A B).
#   TESTS="tests/test_gpu_backward_kernels.py" TAG=r03w bash tools/gpu_r03_trainab.sh "X=1" "MGU_NO_WGRAD_X3=1"
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r03w}
if [ -n "${TESTS:-}" ]; then
  timeout -k 10 900 python -m pytest $TESTS -q -m gpu -p no:cacheprovider --maxfail=20 ${PYTEST_ARGS:-} > gpurun_out/${TAG}_tests.log 2>&1
  echo "tests rc=$?"; tail -n 12 gpurun_out/${TAG}_tests.log | cut -c1-300
fi
for rep in 1 2; do
  i=0
  for env in "${@:-X=1}"; do
    i=$((i+1))
    env $env timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > gpurun_out/${TAG}_train_${i}_$rep.log 2>&1
    echo "== train [$env] rep $rep rc=$?"
    python - <<PY
import json
try:
    j=json.loads([l for l in open("gpurun_out/${TAG}_train_${i}_$rep.log") if l.startswith("{")][-1])
    print("  ", j["value"], "Mpix/s", j["ms_per_step"], "ms", j.get("spread"))
    if $rep == 1:
        for k in (j.get("roofline") or {}).get("kernels", []): print("      ", k["kernel"][:50], k["launches_per_step"], k["ms_per_step"], k["avg_launch_us"])
except Exception as e:
    print("no json", e); print(open("gpurun_out/${TAG}_train_${i}_$rep.log").read()[-1500:])
PY
  done
done
