#!/usr/bin/env bash
# Per-layer report of the headline bench (rocprofv3 kernel trace of one warm run) under optional environment switches.
#   TAG=r02z bash tools/gpu_layers.sh [VAR=VALUE ...]   (BENCH_ARGS="--dtype bf16" for the bf16 mode)
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-layers}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-profile-pass --spread-windows 0 ${BENCH_ARGS:-} > gpurun_out/${TAG}_prof.log 2>&1 || { echo "trace failed"; tail -20 gpurun_out/${TAG}_prof.log; exit 1; }
python tools/layer_report.py gpurun_out/prof_$TAG > gpurun_out/${TAG}_layer_report.txt; cat gpurun_out/${TAG}_layer_report.txt
rm -rf gpurun_out/prof_$TAG
