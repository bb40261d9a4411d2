#!/bin/bash
# tools/gpu_pmc.sh <tag> [bench args] — rocprofv3 counter passes of the bench command on the GPU box, one counter group per pass
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass; SQ block has 8 slots), kernel-trace only next to --pmc.
# Summaries: gpurun_out/<tag>/pmc_{fetch,write,sq1,sq2}.md (+ .json) and pmc_traffic.json (what bench.py attaches as roofline.traffic).
TAG=${1:-pmc}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --repeat-to-seconds 0 --steps 3 --warmup 1 $*"
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_$name -o p -- $BENCH > $O/pmc_$name.log 2>&1
  (cd $R && python tools/pmc_summary.py /tmp/pmc_$name $O/pmc_$name.md > /dev/null 2>>$O/pmc_$name.log)
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R
python - <<PY
import hashlib, json, os
f = json.load(open("$O/pmc_fetch.json")); w = json.load(open("$O/pmc_write.json"))
s1 = json.load(open("$O/pmc_sq1.json")); s2 = json.load(open("$O/pmc_sq2.json"))
args = "$*".split()
lib = os.environ.get("ALOAM_MI355X_LIB", "$R/a-loam_amd/lib/libaloam_mi355x.so")
sq = {k: dict({c: v[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES") if c in v},
              **{c: s2.get(k, {}).get(c) for c in ("SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY") if s2.get(k, {}).get(c) is not None},
              avg_us=s2.get(k, {}).get("avg_us", v.get("avg_us"))) for k, v in s1.items()}
import re
default_batch = int(re.search(r'"--batch", type=int, default=(\d+)', open("$R/bench.py").read()).group(1))
json.dump({"batch": int(args[args.index("--batch") + 1]) if "--batch" in args else default_batch, "mapping": "--mapping" in args,
           "sensor": args[args.index("--sensor") + 1] if "--sensor" in args else "HDL-64",
           "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / two SQ groups, separate passes of python bench.py --no-cpu-baseline --steps 3 --warmup 1 $* (tools/gpu_pmc.sh $TAG)",
           "fetch_kib": {k: v["FETCH_SIZE"] for k, v in f.items() if "FETCH_SIZE" in v}, "write_kib": {k: v["WRITE_SIZE"] for k, v in w.items() if "WRITE_SIZE" in v},
           "avg_us": {k: v.get("avg_us") for k, v in f.items()}, "sq": sq},
          open("$O/pmc_traffic.json", "w"), indent=1)
PY
cat $O/pmc_fetch.md $O/pmc_write.md $O/pmc_sq1.md $O/pmc_sq2.md
