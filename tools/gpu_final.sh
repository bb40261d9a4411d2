#!/bin/bash
# tools/gpu_final.sh <tag> — one call on the GPU box, most important first (the call may be cut short by the GPU-minute budget):
#   1. parity suite (-m gpu) with the product library; if it fails, whatever A/B builds sit in a-loam_amd/lib/variants (built with
#      tools/build_variant.sh, e.g. one change switched off each; names below) are tried on the registration subset to name the change
#      that breaks it, and the first that passes is the library the rest of the script measures
#   2. A/B of k_ring_features: the product library against the build of the previous commit (variants/libold.so), same box
#   3. rocprofv3 kernel stats of the headline workload, SQ instruction counters of k_ring_features
#   4. the default bench line
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.log; }
timeout 300 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; RC=$?
tail -3 $O/pytest_gpu.log; stamp "pytest product rc=$RC"
CHOSEN=product
if [ $RC -ne 0 ]; then
  CHOSEN=""
  SUB='free_running or long_ring or goldens or edge_cases or full_size'
  for v in pick1 noreach notiles nobox selonly reachonly old3; do
    [ -f $R/a-loam_amd/lib/variants/lib$v.so ] || continue
    ALOAM_MI355X_LIB=$R/a-loam_amd/lib/variants/lib$v.so timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k "$SUB" > $O/parity_$v.log 2>&1
    rc=$?; stamp "parity subset $v rc=$rc"
    if [ $rc -eq 0 ] && [ -z "$CHOSEN" ]; then CHOSEN=$v; fi
  done
  grep -n "Error\|assert\|FAILED" $O/pytest_gpu.log | head -20
fi
echo "$CHOSEN" > $O/chosen.txt
stamp "chosen=$CHOSEN"
[ -z "$CHOSEN" ] && exit 1
if [ "$CHOSEN" != product ]; then
  export ALOAM_MI355X_LIB=$R/a-loam_amd/lib/variants/lib$CHOSEN.so
  timeout 300 python -m pytest tests -m gpu -q > $O/pytest_gpu_$CHOSEN.log 2>&1; stamp "pytest $CHOSEN rc=$?"; tail -3 $O/pytest_gpu_$CHOSEN.log
fi
cd /tmp && export TMPDIR=/tmp
ab() {  # name, lib ("" = whatever ALOAM_MI355X_LIB says)
  ( [ -n "$2" ] && export ALOAM_MI355X_LIB=$2
    python $R/bench.py --no-cpu-baseline --no-extras --steps 20 > $O/ab_$1.log 2>&1
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
ab chosen ""
ab old $R/a-loam_amd/lib/variants/libold.so
stamp "A/B done"
rocprofv3 --kernel-trace --stats -d $O/stats_headline -o s -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 > $O/stats_headline.log 2>&1
(cd $R && python tools/rocprof_summary.py $O/stats_headline/s_results.db $O/kernel_stats_headline.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --steps 10" > /dev/null)
rm -rf $O/stats_headline
stamp "kernel stats done"
rm -rf /tmp/pmc_sq1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex k_ring_features --output-format csv -d /tmp/pmc_sq1 -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/pmc_sq1.log 2>&1
(cd $R && python tools/pmc_summary.py /tmp/pmc_sq1 $O/pmc_sq1_ring_features.md 2>>$O/pmc_sq1.log | tail -n +4)
stamp "SQ counters done"
if [ "$CHOSEN" = product ] && [ $(( $(date +%s) - T0 )) -lt 300 ]; then   # time permitting: the radix sort of the run keys, now that the selection is cheaper
  ( cd $R && ALOAM_MI355X_LIB=$R/a-loam_amd/lib/variants/libradix.so timeout 120 python -m pytest tests/test_gpu_parity.py -q -x -k "free_running or long_ring or goldens" > $O/parity_radix.log 2>&1; stamp "parity subset radix rc=$?" )
  ab radix $R/a-loam_amd/lib/variants/libradix.so
fi
cd $R
python bench.py > $O/bench.log 2>&1; stamp "default bench rc=$?"
tail -c 400 $O/bench.log
