#!/bin/bash
# tools/gpu_map_ab.sh <tag> [variant names...] — one call on the GPU box for the mapping stage: the mapping parity tests with the product
# library (unless SKIP_PYTEST), then tools/ab_check.py --mapping with the product library and every named A/B build of
# a-loam_amd/lib/variants (bitwise comparison of the recorded outputs against the product run, per-kernel times side by side).
TAG=${1:-mapab}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.log; }
if [ -z "$SKIP_PYTEST" ]; then
  timeout 600 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-mapping}" > $O/pytest_gpu.log 2>&1; RC=$?
  tail -3 $O/pytest_gpu.log; stamp "pytest rc=$RC"
  [ $RC -ne 0 ] && grep -n "Error\|assert\|FAILED" $O/pytest_gpu.log | head -30
fi
timeout 120 python tools/ab_check.py run product $O/product.npz --mapping --steps ${AB_STEPS:-10} 2>&1 | tail -4 | tee -a $O/timeline.log
for v in "$@"; do
  timeout 120 python tools/ab_check.py run a-loam_amd/lib/variants/lib$v.so $O/$v.npz --mapping --steps ${AB_STEPS:-10} 2>&1 | tail -4 | tee -a $O/timeline.log
  python tools/ab_check.py compare $O/product.npz $O/$v.npz 2>&1 | grep -i "identical\|map_\|sum" | tee -a $O/timeline.log
done
rm -f $O/*.npz
stamp "done"
