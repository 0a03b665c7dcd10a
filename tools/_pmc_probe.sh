#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
rm -rf gpurun_out/pmc_probe
timeout -k 5 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_probe -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --spread-windows 0 --sustained-seconds 0 > gpurun_out/pmc_probe.log 2>&1
echo "probe rc=$?"
tail -3 gpurun_out/pmc_probe.log | cut -c1-200
ls gpurun_out/pmc_probe/*/ 2>/dev/null | head
