#!/usr/bin/env bash
# Stall-reason counters for the conv kernels (two PMC passes, each its own run, --kernel-trace only).
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --spread-windows 0 ${BENCH_ARGS:-}"
run() { local name=$1; shift; rm -rf gpurun_out/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python $ARGS > gpurun_out/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 gpurun_out/pmc_$name.log; exit 1; }; echo "$name ok"; }
run s1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run s2 SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run s3 SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY
