#!/usr/bin/env bash
# per-layer report of the forward (rocprofv3 kernel trace of bench.py): DTYPE=f32|bf16
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r04l}
DT=${DTYPE:-f32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$DT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$DT -- python bench.py --steps 20 --warmup 5 --dtype $DT --no-cpu-baseline --no-profile-pass --spread-windows 0 --sustained-seconds 0 > gpurun_out/${TAG}_prof_$DT.log 2>&1 || { echo "prof failed"; tail -20 gpurun_out/${TAG}_prof_$DT.log; exit 1; }
python tools/layer_report.py gpurun_out/prof_$DT > gpurun_out/${TAG}_layer_report_$DT.txt; cat gpurun_out/${TAG}_layer_report_$DT.txt
rm -rf gpurun_out/prof_$DT
