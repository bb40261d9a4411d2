#!/bin/bash
# tools/gpu_pmc_sq.sh <tag> [bench args] — SQ-level counters of every aloam:: kernel (one --pmc pass), summary to gpurun_out/<tag>_pmc_sq.md
TAG=${1:-sq}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_sq -o p -- python $R/bench.py --no-cpu-baseline --batch 64 --steps 3 --warmup 1 --frames 3 "$@" > /tmp/pmc_sq.log 2>&1
cd $R && python tools/pmc_summary.py /tmp/pmc_sq gpurun_out/${TAG}_pmc_sq.md
