#!/bin/bash
# tools/gpu_variants_map.sh <tag> <variants...> — like gpu_variants.sh for the configs[2] workload (bench.py --mapping, 8 steps)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = product ]; then unset ALOAM_MI355X_LIB; else export ALOAM_MI355X_LIB=$R/a-loam_amd/lib/variants/lib$v.so; fi
  python $R/bench.py --no-cpu-baseline --no-extras --mapping --steps 8 > $O/benchmap_$v.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("$O/benchmap_$v.log").read().strip().splitlines()[-1])
    k = d["roofline"]["kernels_ms_per_step"]
    print("$v", d["ms_per_step"], json.dumps({a: b for a, b in k.items() if a.startswith("map")}))
except Exception as e:
    print("$v FAILED", e); print(open("$O/benchmap_$v.log").read()[-1500:])
PY
done
