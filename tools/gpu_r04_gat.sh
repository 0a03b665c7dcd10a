#!/usr/bin/env bash
# Round-4 GAT session: the graph-layer tests, then tools/gat_ab.py under each environment variant given as an argument.
#   TAG=r04g bash tools/gpu_r04_gat.sh "MGU_GAT_FUSED_V=1" "MGU_GAT_FUSED_V=2"
set -u
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r04g}
TESTS=${TESTS:-"tests/test_gpu_gat_schedules.py tests/test_gpu_gat_train.py tests/test_gpu_region.py tests/test_gpu_mincut.py tests/test_gpu_parity.py"}
if [ "$TESTS" != "none" ]; then
  timeout -k 10 900 python -m pytest $TESTS -q -m gpu -p no:cacheprovider --maxfail=10 > gpurun_out/${TAG}_tests.log 2>&1
  echo "tests rc=$?"; tail -n 12 gpurun_out/${TAG}_tests.log | cut -c1-300
fi
i=0
for env in "${@:-X=1}"; do
  i=$((i+1))
  echo "== gat [$env]"
  env $env timeout -k 10 300 python tools/gat_ab.py > gpurun_out/${TAG}_gat_$i.log 2>&1 || tail -n 20 gpurun_out/${TAG}_gat_$i.log
  grep -E "graphs_|c4_" gpurun_out/${TAG}_gat_$i.log | cut -c1-260
done
