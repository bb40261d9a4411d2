#!/bin/bash
# tools/gpu_round.sh <tag> [notests] — one GPU-box visit: parity tests, default bench (with cpu_baseline), rocprofv3 kernel
# stats of the same bench command, and two separate --pmc passes (FETCH_SIZE / WRITE_SIZE) for the roofline traffic.
# Outputs land in gpurun_out/<tag>/ ; copy the summaries to profiles/ afterwards.
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" != "notests" ]; then python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; fi
python bench.py > $O/bench.log 2>&1; tail -c 2500 $O/bench.log
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $BENCH > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_fetch -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_write -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_write.log 2>&1
cd $R
python tools/rocprof_summary.py $O/stats/s_results.db $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline" > /dev/null
python tools/pmc_summary.py /tmp/pmc_fetch $O/pmc_fetch.md > /dev/null
python tools/pmc_summary.py /tmp/pmc_write $O/pmc_write.md > /dev/null
rm -rf $O/stats
python - <<PY
import json
f = json.load(open("$O/pmc_fetch.json")); w = json.load(open("$O/pmc_write.json"))
b = json.loads(open("$O/bench.log").read().strip().splitlines()[-1])
json.dump({"batch": b["config"]["sequences_per_gpu"], "mapping": False, "sensor": "HDL-64", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of python bench.py --no-cpu-baseline --steps 4 --warmup 1 (tools/gpu_round.sh $TAG)",
           "fetch_kib": {k: v["FETCH_SIZE"] for k, v in f.items() if "FETCH_SIZE" in v}, "write_kib": {k: v["WRITE_SIZE"] for k, v in w.items() if "WRITE_SIZE" in v}},
          open("$O/pmc_traffic.json", "w"), indent=1)
PY
cat $O/kernel_stats.md $O/pmc_fetch.md $O/pmc_write.md
