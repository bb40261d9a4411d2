#!/bin/bash
# tools/gpu_map_trace.sh <tag> <product|variant names...> — per-kernel dispatch times of the mapping stage (rocprofv3 --kernel-trace over
# tools/ab_check.py --mapping, torch-free): median / mean over the dispatches that did work, per library.  KREGEX selects the kernels.
TAG=${1:-maptrace}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=product; [ "$v" != product ] && lib=$R/a-loam_amd/lib/variants/lib$v.so
  rm -rf /tmp/kt_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -o k -- python $R/tools/ab_check.py run $lib /tmp/kt_$v.npz --mapping --steps ${AB_STEPS:-6} > $O/trace_$v.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import csv, glob, re, statistics as st
rows = []
for f in glob.glob("/tmp/kt_$v/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aloam::", "")
    if not re.search(r"${KREGEX:-map_|vox}", n): continue
    acc.setdefault(n, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
acc = {n: [x for _, x in sorted(v)] for n, v in acc.items()}
print("== $v")
for n, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w = [x for x in d if x > 0.25 * max(d)]
    print(f"{n:44s} n {len(d):3d} worked {len(w):3d} median {st.median(w):8.1f} mean {st.mean(w):8.1f} max {max(d):8.1f} us  total {sum(d)/1000:7.2f} ms")
    if "${SHOW_ALL:-}": print("      in launch order:", " ".join(f"{x:.0f}" for x in d))
PY
  rm -rf /tmp/kt_$v /tmp/kt_$v.npz
done
