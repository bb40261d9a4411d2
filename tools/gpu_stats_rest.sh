#!/bin/bash
# tools/gpu_stats_rest.sh <tag> — rocprofv3 kernel stats of configs[2] (--mapping) and configs[3] (--sensor ROWS128) and the second SQ
# counter group of k_ring_features, for a binary whose headline evidence tools/gpu_final.sh already took.
TAG=${1:-rest}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "mapping:--mapping" "rows128:--sensor ROWS128"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d $O/stats_$name -o s -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 $args > $O/stats_$name.log 2>&1
  (cd $R && python tools/rocprof_summary.py $O/stats_$name/s_results.db $O/kernel_stats_$name.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --steps 10 $args" > /dev/null)
  rm -rf $O/stats_$name
  tail -c 300 $O/stats_$name.log
done
rm -rf /tmp/pmc_sq2
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --kernel-include-regex k_ring_features --output-format csv -d /tmp/pmc_sq2 -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/pmc_sq2.log 2>&1
(cd $R && python tools/pmc_summary.py /tmp/pmc_sq2 $O/pmc_sq2_ring_features.md 2>>$O/pmc_sq2.log | tail -n +4)
