#!/usr/bin/env bash
# A/B of an environment switch: kernel + parity tests under the switch, then the headline bench with and without it.
# usage: tools/gpu_ab.sh VAR=VALUE   (e.g. MGU_WINO_PREC=1)
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
SW="${1:?VAR=VALUE}"
env "$SW" timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider > gpurun_out/kernels.log 2>&1 || { tail -50 gpurun_out/kernels.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/kernels.log
env "$SW" timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > gpurun_out/parity.log 2>&1 || { tail -50 gpurun_out/parity.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/parity.log
env "$SW" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1 || { tail -20 gpurun_out/bench_a.log; exit 1; }
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"max_abs_logit_err[a-z_]*": [0-9.e-]*' gpurun_out/bench_a.log | tr '\n' ' '; echo " ($SW)"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_b.log 2>&1 || { tail -20 gpurun_out/bench_b.log; exit 1; }
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"max_abs_logit_err[a-z_]*": [0-9.e-]*' gpurun_out/bench_b.log | tr '\n' ' '; echo " (default)"
echo done
