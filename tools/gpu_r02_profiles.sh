#!/usr/bin/env bash
# Round-2 profile session: warm rocprofv3 kernel stats of the three bench configurations + FETCH/WRITE PMC passes (fp32 and bf16).
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r02a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() { # name warm steps args...
  local name=$1 warm=$2 steps=$3; shift 3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-profile-pass --spread-windows 0 "$@" > gpurun_out/${TAG}_prof_$name.log 2>&1 || { echo "prof $name failed"; tail -20 gpurun_out/${TAG}_prof_$name.log; return 1; }
  python tools/warm_kernel_stats.py gpurun_out/prof_$name $warm $steps gpurun_out/${TAG}_${name}_warm_kernel_stats.csv
  tail -n 1 gpurun_out/${TAG}_prof_$name.log | cut -c1-300
}
prof f32 5 20 && prof bf16 5 20 --dtype bf16 && prof train 3 10 --mode train || exit 1
pmc() { # tag args...
  local tag=$1; shift
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    local n=${c%%:*} ctr=${c##*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/pmc_$n -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --spread-windows 0 "$@" > gpurun_out/${TAG}_pmc_${tag}_$n.log 2>&1 || { echo "pmc $tag $n failed"; tail -20 gpurun_out/${TAG}_pmc_${tag}_$n.log; return 1; }
  done
  python tools/traffic_kernels.py gpurun_out/pmc_${tag} gpurun_out/${TAG}_traffic_${tag}.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass $*"
}
pmc f32 && pmc bf16 --dtype bf16 && pmc train --mode train
rm -rf gpurun_out/prof_f32 gpurun_out/prof_bf16 gpurun_out/prof_train gpurun_out/pmc_f32 gpurun_out/pmc_bf16 gpurun_out/pmc_train
echo "=== profiles done"
