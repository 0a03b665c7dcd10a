#!/usr/bin/env bash
# One GPU-box session: kernel tests -> parity tests -> smoke -> bench (-> optional rocprof).
# Stops at the first failing step; every step writes its log under gpurun_out/.
set -u
mkdir -p gpurun_out
step() {  # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name"
  if ! timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; then
    echo "!!! $name failed (rc=$?)"; tail -n 60 "gpurun_out/$name.log"; exit 1
  fi
  tail -n "${TAILN:-6}" "gpurun_out/$name.log"
}
export PYTHONDONTWRITEBYTECODE=1
step kernels 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider
step parity 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -p no:cacheprovider
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py --steps 20 --warmup 5
if [ "${PROFILE:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass
  ls -R gpurun_out/prof | head -20
fi
echo "=== all steps ok"
