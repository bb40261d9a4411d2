#!/bin/bash
# tools/gpu_evidence.sh <tag> — the evidence set of a round on the GPU box: parity tests, the default bench line, rocprofv3 kernel
# stats of the same command, and the counter passes (FETCH_SIZE / WRITE_SIZE / two SQ groups) for the headline workload, plus
# FETCH / WRITE for configs[2] (--mapping) and configs[3] (--sensor ROWS128).  Everything lands in gpurun_out/<tag>/.
TAG=${1:-evidence}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench.log 2>&1; tail -c 600 $O/bench.log
cd /tmp && export TMPDIR=/tmp
for cfg in "headline:" "mapping:--mapping" "rows128:--sensor ROWS128 --batch 1024" "travel:--travel --frames 24 --batch 1024"; do   # the comparison workloads keep the batch of the earlier rounds
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d $O/stats_$name -o s -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 $args > $O/stats_$name.log 2>&1
  (cd $R && python tools/rocprof_summary.py $O/stats_$name/s_results.db $O/kernel_stats_$name.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --steps 10 $args" > /dev/null)
  rm -rf $O/stats_$name
done
cd $R
bash tools/gpu_pmc.sh $TAG/pmc_headline --no-extras > /dev/null 2>&1
for cfg in "mapping:--mapping" "rows128:--sensor ROWS128 --batch 1024"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_${name}_$c
      rocprofv3 --pmc $c --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_${name}_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 $args > $O/pmc_${name}_$c.log 2>&1
      (cd $R && PMC_LAST=$([ $name = mapping ] && echo 6 || echo 0) python tools/pmc_summary.py /tmp/pmc_${name}_$c $O/pmc_${name}_$c.md > /dev/null 2>&1)
    done )
done
# configs[2] at steady-state map depth (--mapping = the travelling workload, 256 sequences): instruction counters and the L1 / texture-addresser view of the submap search
( cd /tmp
  rm -rf /tmp/pmc_mapping_sq1 /tmp/pmc_mapping_tcp
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_mapping_sq1 -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --mapping > $O/pmc_mapping_sq1.log 2>&1
  (cd $R && PMC_LAST=6 python tools/pmc_summary.py /tmp/pmc_mapping_sq1 $O/pmc_mapping_sq1.md > /dev/null 2>&1)
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "map_search|mapgrid|vox_lds" --output-format csv -d /tmp/pmc_mapping_tcp -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --mapping > $O/pmc_mapping_tcp.log 2>&1
  (cd $R && PMC_LAST=6 python tools/pmc_summary.py /tmp/pmc_mapping_tcp $O/pmc_mapping_tcp.md > /dev/null 2>&1) )
ls $O
