#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kind in patch stress; do
  G=64; [ "$kind" = stress ] && G=32
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gat_${kind}_trace -- python tools/gat_bench.py --graphs $G --kind $kind > gpurun_out/gat_${kind}.log 2>&1 || { tail gpurun_out/gat_${kind}.log; exit 1; }
  grep graphs= gpurun_out/gat_${kind}.log
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/gat_${kind}_fetch -- python tools/gat_bench.py --graphs $G --kind $kind --iters 3 > /dev/null 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/gat_${kind}_write -- python tools/gat_bench.py --graphs $G --kind $kind --iters 3 > /dev/null 2>&1 || exit 1
done
echo done
