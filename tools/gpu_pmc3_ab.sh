#!/usr/bin/env bash
# Stall-reason counters (tools/gpu_pmc3.sh) for two environment variants; reports per wino kernel family.
set -u
mkdir -p gpurun_out
TAG=${TAG:-r02}
i=0
for env in "$@"; do
  i=$((i+1))
  env $env bash tools/gpu_pmc3.sh > gpurun_out/${TAG}_pmc3_$i.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc3_$i.log; exit 1; }
  { echo "# $env"; python tools/pmc3_report.py gpurun_out ${PAT:-wino}; python tools/pmc3_calib.py gpurun_out "${PAT:-wino3x3}" 2.3; } > gpurun_out/${TAG}_pmc3_report_$i.txt 2>&1
  rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2 gpurun_out/pmc_s3
  echo "== $env"; cat gpurun_out/${TAG}_pmc3_report_$i.txt | head -120
done
