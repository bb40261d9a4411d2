#!/bin/bash
# tools/gpu_pmc_ab.sh <tag> <kernel regex> [variant names...] — SQ instruction counters (one pass) of some kernels for the product library and A/B builds
TAG=$1; RX=$2; shift; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { local name=$1; rm -rf /tmp/pmc_$name
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc_$name -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 $BENCH_ARGS > $O/pmc_$name.log 2>&1
  echo "== $name"; (cd $R && python tools/pmc_summary.py /tmp/pmc_$name $O/pmc_$name.md 2>>$O/pmc_$name.log | tail -n +4); }
one product
for v in "$@"; do ( export ALOAM_MI355X_LIB=$R/a-loam_amd/lib/variants/lib$v.so; one $v ); done
