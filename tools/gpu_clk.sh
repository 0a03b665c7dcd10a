#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/pmc_clk.log 2>&1 || { tail -20 gpurun_out/pmc_clk.log; exit 1; }
echo ok
