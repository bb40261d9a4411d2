#!/bin/bash
# tools/gpu_map_pmc.sh <tag> "<counters>" <product|variant...> — one counter pass (rocprofv3 --pmc, its own run: no other trace domain)
# over tools/ab_check.py --mapping for the kernels matching KREGEX; prints every dispatch (time + counters) in launch order.
TAG=$1; CNT=$2; shift; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=product; [ "$v" != product ] && lib=$R/a-loam_amd/lib/variants/lib$v.so
  rm -rf /tmp/pm_$v
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace --kernel-include-regex "${KREGEX:-map_search}" --output-format csv -d /tmp/pm_$v -o p -- python $R/tools/ab_check.py run $lib /tmp/pm_$v.npz ${AB_MAPPING---mapping} --steps ${AB_STEPS:-4} > $O/pmc_$v.log 2>&1
  python - <<PY | tee -a $O/pmc_summary.txt
import glob, pandas as pd
cc = pd.concat([pd.read_csv(f) for f in glob.glob("/tmp/pm_$v/**/*counter_collection.csv", recursive=True)])
kt = pd.concat([pd.read_csv(f) for f in glob.glob("/tmp/pm_$v/**/*kernel_trace.csv", recursive=True)])
cc["kernel"] = cc.Kernel_Name.str.extract(r"aloam::(\w+(?:<[^>]*>)?)")
piv = cc.pivot_table(index=["Dispatch_Id", "kernel"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
kt["us"] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
piv = piv.merge(kt[["Dispatch_Id", "us"]], on="Dispatch_Id").sort_values("Dispatch_Id")
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30); pd.set_option("display.float_format", lambda x: f"{x:,.0f}")
print("== $v"); print(piv.drop(columns=["Dispatch_Id"]).to_string(index=False))
PY
  rm -rf /tmp/pm_$v /tmp/pm_$v.npz
done
