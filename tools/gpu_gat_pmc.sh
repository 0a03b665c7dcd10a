#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/gat_pmc -- python tools/gat_bench.py --graphs 64 --kind patch --iters 3 > gpurun_out/gat_pmc.log 2>&1 || { tail gpurun_out/gat_pmc.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/gat_pmc2 -- python tools/gat_bench.py --graphs 64 --kind patch --iters 3 > gpurun_out/gat_pmc2.log 2>&1 || { tail gpurun_out/gat_pmc2.log; exit 1; }
echo ok
