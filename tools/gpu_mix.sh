#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_mix -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/pmc_mix.log 2>&1 || { tail -20 gpurun_out/pmc_mix.log; exit 1; }
echo ok
