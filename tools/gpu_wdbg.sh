#!/usr/bin/env bash
# Diagnosis: Winograd kernel with parts disabled (MGU_WINO_DBG), per-layer timings from a kernel trace each.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
for v in ${DBGS:-0 1 2 4 7}; do
  export MGU_WINO_DBG=$v
  rm -rf gpurun_out/prof_dbg$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_dbg$v -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/dbg$v.log 2>&1 || { tail -5 gpurun_out/dbg$v.log; exit 1; }
  echo "dbg $v done"
done
