#!/usr/bin/env bash
# Round-5 profile session: bench lines of the four configurations, warm rocprofv3 kernel stats, per-layer report, FETCH/WRITE PMC
# passes (fp32 / bf16 / train) -> gpurun_out/<TAG>_*; the summaries to keep are copied into profiles/ by hand.
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r05p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2.json.log 2>&1; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_c3prec_bf16.json.log 2>&1; echo "bench bf16 rc=$?"
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_train_c5.json.log 2>&1; echo "bench train rc=$?"
timeout -k 10 600 python bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_c4.json.log 2>&1; echo "bench c4 rc=$?"
# box calibration: boxes of the pool differ by +-4 %; the same step on the C++ kernels (unchanged since round 4) beside the assembly kernels
timeout -k 10 300 python tools/ab_env_step.py 3 - MGU_WINO_ASM=0 > gpurun_out/${TAG}_ab_asm_cpp.txt 2>&1; echo "ab rc=$?"
prof() { # name warm steps args...
  local name=$1 warm=$2 steps=$3; shift 3
  rm -rf gpurun_out/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-profile-pass --spread-windows 0 --sustained-seconds 0 "$@" > gpurun_out/${TAG}_prof_$name.log 2>&1 || { echo "prof $name failed"; tail -20 gpurun_out/${TAG}_prof_$name.log; return 1; }
  python tools/warm_kernel_stats.py gpurun_out/prof_$name $warm $steps gpurun_out/${TAG}_bench_${name}_warm_kernel_stats.csv | head -4
}
prof f32 5 20 && python tools/layer_report.py gpurun_out/prof_f32 > gpurun_out/${TAG}_layer_report.txt
prof bf16 5 20 --dtype bf16
prof train 3 10 --mode train && python tools/step_sequence.py gpurun_out/prof_train > gpurun_out/${TAG}_train_step_sequence.txt
prof c4 2 5 --workload c4
pmc() { # tag args...
  local tag=$1; shift
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    local n=${c%%:*} ctr=${c##*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/pmc_$n -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --spread-windows 0 --sustained-seconds 0 "$@" > gpurun_out/${TAG}_pmc_${tag}_$n.log 2>&1 || { echo "pmc $tag $n failed"; tail -20 gpurun_out/${TAG}_pmc_${tag}_$n.log; return 1; }
  done
  python tools/traffic_kernels.py gpurun_out/pmc_${tag} gpurun_out/${TAG}_traffic_${tag}.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass $*"
}
pmc f32 && pmc bf16 --dtype bf16 && pmc train --mode train
rm -rf gpurun_out/prof_f32 gpurun_out/prof_bf16 gpurun_out/prof_train gpurun_out/prof_c4 gpurun_out/pmc_f32 gpurun_out/pmc_bf16 gpurun_out/pmc_train
python tools/gat_ab.py > gpurun_out/${TAG}_gat_layers.txt 2>&1; grep -E "graphs_|c4_" gpurun_out/${TAG}_gat_layers.txt | cut -c1-200
python tools/gat_tiles_sweep.py > gpurun_out/${TAG}_gat_tiles_sweep.txt 2>&1; grep graphs gpurun_out/${TAG}_gat_tiles_sweep.txt | tail -3
echo "=== profiles done"
