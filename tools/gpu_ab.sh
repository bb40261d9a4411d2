#!/bin/bash
# tools/gpu_ab.sh <tag> [variant names...] — one call on the GPU box: the parity suite with the product library, then the headline bench
# (per-kernel hipEvent times) with the product library and with every named A/B build of a-loam_amd/lib/variants, same box, same run.
TAG=${1:-ab}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.log; }
if [ -z "$SKIP_PYTEST" ]; then
  timeout 400 python -m pytest tests -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} > $O/pytest_gpu.log 2>&1; RC=$?
  tail -3 $O/pytest_gpu.log; stamp "pytest product rc=$RC"
  [ $RC -ne 0 ] && grep -n "Error\|assert\|FAILED" $O/pytest_gpu.log | head -30
fi
cd /tmp && export TMPDIR=/tmp
ab() {  # name, lib
  ( [ -n "$2" ] && export ALOAM_MI355X_LIB=$2
    python $R/bench.py --no-cpu-baseline --no-extras --steps ${AB_STEPS:-40} > $O/ab_$1.log 2>&1
    python - <<PY
import json
try:
    d = json.loads(open("$O/ab_$1.log").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], json.dumps(d["roofline"]["kernels_ms_per_step"]))
except Exception as e:
    print("$1 FAILED", e); print(open("$O/ab_$1.log").read()[-1500:])
PY
  ) | tee -a $O/timeline.log
}
ab product ""
for v in "$@"; do ab $v $R/a-loam_amd/lib/variants/lib$v.so; done
stamp "A/B done"
