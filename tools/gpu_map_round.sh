#!/bin/bash
# tools/gpu_map_round.sh <tag> — bench.py --mapping and the rocprofv3 kernel table of the same command (gpurun_out/<tag>/).
TAG=${1:-maprun}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --mapping --steps 10 > $O/bench_mapping.log 2>&1; tail -c 1800 $O/bench_mapping.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --mapping --steps 10 > $O/stats.log 2>&1
cd $R
python tools/rocprof_summary.py $O/stats/s_results.db $O/kernel_stats_mapping.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --mapping --steps 10" > /dev/null
rm -rf $O/stats
head -40 $O/kernel_stats_mapping.md
