#!/bin/bash
# tools/gpu_variants.sh <tag> <variant names...> — per-kernel hipEvent times (and SQ instruction counters) of A/B builds of the
# library, one short bench run each.  "product" = a-loam_amd/lib/libaloam_mi355x.so.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = product ]; then unset ALOAM_MI355X_LIB; else export ALOAM_MI355X_LIB=$R/a-loam_amd/lib/variants/lib$v.so; fi
  python $R/bench.py --no-cpu-baseline --no-extras --steps 10 > $O/bench_$v.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$v.log").read().strip().splitlines()[-1])
    print("$v", d["ms_per_step"], json.dumps(d["roofline"]["kernels_ms_per_step"]))
except Exception as e:
    print("$v FAILED", e); print(open("$O/bench_$v.log").read()[-1500:])
PY
  if [ -n "$PMC" ]; then
    rm -rf /tmp/pmc_$v
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "$PMC" --output-format csv -d /tmp/pmc_$v -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/pmc_$v.log 2>&1
    (cd $R && python tools/pmc_summary.py /tmp/pmc_$v $O/pmc_$v.md | tail -8)
  fi
done
