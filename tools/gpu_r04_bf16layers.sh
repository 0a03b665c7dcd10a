#!/usr/bin/env bash
# per-layer report of the bf16 forward (rocprofv3 kernel trace of bench.py --dtype bf16)
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r04bf}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_bf16
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bf16 -- python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-profile-pass --spread-windows 0 --sustained-seconds 0 > gpurun_out/${TAG}_prof.log 2>&1 || { echo "prof failed"; tail -20 gpurun_out/${TAG}_prof.log; exit 1; }
python tools/layer_report.py gpurun_out/prof_bf16 > gpurun_out/${TAG}_layer_report.txt; cat gpurun_out/${TAG}_layer_report.txt
rm -rf gpurun_out/prof_bf16
