#!/usr/bin/env bash
# One PMC pass with clock / MFMA-busy counters (own run, --kernel-trace only).
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
rm -rf gpurun_out/pmc_clk
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass ${BENCH_ARGS:-} > gpurun_out/pmc_clk.log 2>&1 || { tail -20 gpurun_out/pmc_clk.log; exit 1; }
echo done
