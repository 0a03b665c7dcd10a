#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > gpurun_out/tests_ab.log 2>&1 || { tail -50 gpurun_out/tests_ab.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/tests_ab.log
for v in "MGU_HALO_TPS1=1" "MGU_HALO_TPS1=0"; do
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$v.log 2>&1 || { tail -20 gpurun_out/bench_$v.log; exit 1; }
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/bench_$v.log | tr '\n' ' '; echo " ($v)"
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tps -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > gpurun_out/rocprof_tps.log 2>&1 || { tail -20 gpurun_out/rocprof_tps.log; exit 1; }
echo done
