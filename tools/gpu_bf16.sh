#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1 || { tail -40 gpurun_out/tests_gpu.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/tests_gpu.log
timeout -k 10 300 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1 || { tail -20 gpurun_out/bench_bf16.log; exit 1; }
tail -1 gpurun_out/bench_bf16.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bf16 -- python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > gpurun_out/rocprof_bf16.log 2>&1 || { tail -20 gpurun_out/rocprof_bf16.log; exit 1; }
echo done
