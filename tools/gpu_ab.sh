#!/usr/bin/env bash
# kernel tests + parity, then A/B bench: generic gather kernel vs halo kernel
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider > gpurun_out/kernels.log 2>&1 || { tail -50 gpurun_out/kernels.log; exit 1; }
tail -2 gpurun_out/kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > gpurun_out/parity.log 2>&1 || { tail -50 gpurun_out/parity.log; exit 1; }
tail -2 gpurun_out/parity.log
MGU_NO_HALO=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_nohalo.log 2>&1 || { tail -20 gpurun_out/bench_nohalo.log; exit 1; }
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/bench_nohalo.log | tr '\n' ' '; echo " (generic)"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_halo.log 2>&1 || { tail -20 gpurun_out/bench_halo.log; exit 1; }
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/bench_halo.log | tr '\n' ' '; echo " (halo)"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_halo -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > gpurun_out/rocprof_halo.log 2>&1 || { tail -20 gpurun_out/rocprof_halo.log; exit 1; }
echo done
