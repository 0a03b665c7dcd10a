#!/usr/bin/env bash
# After gpu_r05_profiles.sh's traffic JSONs (stamped with the library's build id) have been copied into profiles/: the four bench lines again, now
# with roofline.traffic attached, and the fp32 kernel stats + per-layer report from one more trace of the same library -> gpurun_out/<TAG>_*
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r05f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2.json.log 2>&1; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_c3prec_bf16.json.log 2>&1; echo "bench bf16 rc=$?"
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_train_c5.json.log 2>&1; echo "bench train rc=$?"
timeout -k 10 600 python bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_c4.json.log 2>&1; echo "bench c4 rc=$?"
timeout -k 10 300 python tools/ab_env_step.py 3 - MGU_WINO_ASM=0 > gpurun_out/${TAG}_ab_asm_cpp.txt 2>&1; echo "ab rc=$?"
rm -rf gpurun_out/prof_f32
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f32 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile-pass --spread-windows 0 --sustained-seconds 0 > gpurun_out/${TAG}_prof_f32.log 2>&1 || { echo "prof failed"; exit 1; }
python tools/warm_kernel_stats.py gpurun_out/prof_f32 5 20 gpurun_out/${TAG}_bench_f32_warm_kernel_stats.csv | head -4
python tools/layer_report.py gpurun_out/prof_f32 > gpurun_out/${TAG}_layer_report.txt
python tools/step_sequence.py gpurun_out/prof_f32 conv3x3_first_mfma > gpurun_out/${TAG}_forward_step_sequence.txt
rm -rf gpurun_out/prof_f32
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== final lines done"
