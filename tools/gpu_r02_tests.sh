#!/usr/bin/env bash
# Round-2 GPU session: the whole -m gpu suite (one process, no -x so every failure is seen), smoke, both bench modes.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r02a}
timeout -k 10 ${TEST_TO:-900} python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=40 -s ${PYTEST_ARGS:-} > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; tail -n 40 gpurun_out/${TAG}_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 gpurun_out/${TAG}_smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -n 2 gpurun_out/${TAG}_bench.log
MGU_WINO_PREC=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_prec0.log 2>&1; echo "bench prec0 rc=$?"; tail -n 2 gpurun_out/${TAG}_bench_prec0.log
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_train.log 2>&1; echo "train rc=$?"; tail -n 2 gpurun_out/${TAG}_bench_train.log
timeout -k 10 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.log 2>&1; echo "bf16 rc=$?"; tail -n 2 gpurun_out/${TAG}_bench_bf16.log
