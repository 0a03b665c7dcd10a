#!/usr/bin/env bash
# Round-3 baseline session: every -m gpu test (one process), smoke, the headline bench (with the `sustained` record), the 2-rank
# self-launch rehearsal of `bench.py --gpus 2` on one GPU.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r03a}
timeout -k 10 ${TEST_TO:-900} python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=40 ${PYTEST_ARGS:-} > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; tail -n 12 gpurun_out/${TAG}_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/${TAG}_smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/${TAG}_bench.log | cut -c1-900
MGU_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > gpurun_out/${TAG}_bench_2rank.log 2>&1; echo "2-rank rehearsal rc=$?"; tail -n 1 gpurun_out/${TAG}_bench_2rank.log | cut -c1-400
