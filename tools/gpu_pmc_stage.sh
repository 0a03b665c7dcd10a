#!/usr/bin/env bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes, --kernel-trace only) of the "next row" stage benchmarks.
# usage: tools/gpu_pmc_stage.sh  -> gpurun_out/pmcs_{fetch,write}_{ncut,region,det}/
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
run() { # tag counter script args...
  local tag=$1 ctr=$2; shift 2
  rm -rf gpurun_out/pmcs_${ctr}_$tag
  timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmcs_${ctr}_$tag -- python "$@" > gpurun_out/pmcs_${ctr}_$tag.log 2>&1 || { echo "pmc $tag $ctr failed"; tail -5 gpurun_out/pmcs_${ctr}_$tag.log; exit 1; }
  echo "$tag $ctr ok"
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  run ncut $ctr tools/ncut_bench.py --graphs 64 --iters 3 --cpu-graphs 1
  run region $ctr tools/region_bench.py --iters 3
  run det $ctr tools/dethead_bench.py --iters 3
done
echo done
