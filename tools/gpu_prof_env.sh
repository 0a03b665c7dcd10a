#!/usr/bin/env bash
# rocprofv3 kernel trace of the headline bench under environment switches.
# usage: tools/gpu_prof_env.sh OUTDIR VAR=VALUE [VAR=VALUE ...]
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
OUT="gpurun_out/$1"; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > gpurun_out/rocprof_env.log 2>&1 || { tail -20 gpurun_out/rocprof_env.log; exit 1; }
tail -1 gpurun_out/rocprof_env.log | cut -c1-200
