#!/bin/bash
# tools/gpu_pmc_kernel.sh <tag> <kernel regex> [bench args] — SQ counter passes (two groups) restricted to some kernels.
TAG=$1; RX=$2; shift; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 $*"
pass() { local name=$1; shift; rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$name -o p -- $BENCH > $O/pmc_$name.log 2>&1
  (cd $R && python tools/pmc_summary.py /tmp/pmc_$name $O/pmc_$name.md 2>>$O/pmc_$name.log | tail -n +4); }
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
