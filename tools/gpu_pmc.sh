#!/usr/bin/env bash
# PMC passes (each in its own run, --kernel-trace only) over a short bench run.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass ${BENCH_ARGS:-}"
run() { # name counters...
  local name=$1; shift
  echo "=== pmc $name: $*"
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python $ARGS > gpurun_out/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -20 gpurun_out/pmc_$name.log; exit 1; }
  find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -2
}
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32
echo "=== pmc done"
