#!/usr/bin/env bash
# Kernel sequence of ONE warm train step (rocprofv3 --kernel-trace) + warm per-kernel stats.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r03t}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --mode train --steps 5 --warmup 3 --no-cpu-baseline --no-profile-pass --spread-windows 0 > gpurun_out/${TAG}_prof.log 2>&1 || { echo "trace failed"; tail -20 gpurun_out/${TAG}_prof.log; exit 1; }
python tools/step_sequence.py gpurun_out/prof_$TAG > gpurun_out/${TAG}_train_sequence.txt; tail -3 gpurun_out/${TAG}_train_sequence.txt
python tools/warm_kernel_stats.py gpurun_out/prof_$TAG 3 5 gpurun_out/${TAG}_train_warm_kernel_stats.csv | head -3
rm -rf gpurun_out/prof_$TAG
