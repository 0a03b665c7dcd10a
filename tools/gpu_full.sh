#!/usr/bin/env bash
# Full GPU session: every -m gpu test, smoke, headline bench, train bench.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 1000 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1 || { tail -60 gpurun_out/tests_gpu.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/tests_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/bench_train.log 2>&1 || { tail -20 gpurun_out/bench_train.log; exit 1; }
tail -1 gpurun_out/bench_train.log
if [ "${PROFILE:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python bench.py --mode train --steps 3 --warmup 1 > gpurun_out/rocprof_train.log 2>&1 || { tail -20 gpurun_out/rocprof_train.log; exit 1; }
fi
echo "=== all ok"
