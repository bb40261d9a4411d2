"""Summarise rocprofv3 --pmc CSV output (counter_collection + kernel_trace) for the aloam:: kernels.
    python tools/pmc_summary.py <dir-with-csvs> <out.md>"""
import glob, os, sys
import pandas as pd


def main(d, out):
    cc = pd.concat([pd.read_csv(f) for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)])
    cc = cc[cc.Kernel_Name.str.contains("aloam::")]
    cc["kernel"] = cc.Kernel_Name.str.extract(r"aloam::(\w+(?:<[^>]*>)?)")
    last = int(os.environ.get("PMC_LAST", "0"))     # only the last N dispatches of every kernel (a workload that warms up for many steps: the map at its final depth)
    keep = None
    if last > 0:
        ids = cc[["kernel", "Dispatch_Id"]].drop_duplicates().sort_values("Dispatch_Id")
        keep = set(ids.groupby("kernel").tail(last).Dispatch_Id)
        cc = cc[cc.Dispatch_Id.isin(keep)]
    piv = cc.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    cnt = cc.groupby("kernel").Dispatch_Id.nunique().rename("dispatches")
    piv = piv.join(cnt)
    kts = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if kts:
        kt = pd.concat([pd.read_csv(f) for f in kts])
        kt = kt[kt.Kernel_Name.str.contains("aloam::")]
        kt["kernel"] = kt.Kernel_Name.str.extract(r"aloam::(\w+(?:<[^>]*>)?)")
        if keep is not None:
            kt = kt[kt.Dispatch_Id.isin(keep)]
        kt["us"] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
        piv = piv.join(kt.groupby("kernel").us.mean().rename("avg_us"))
        # spread of the dispatch durations, and the two buffer parities apart (even / odd dispatch of a kernel: the last clouds flip every step)
        kt = kt.sort_values("Start_Timestamp")
        kt["parity"] = kt.groupby("kernel").cumcount() % 2
        sp = kt.groupby("kernel").us.agg(min_us="min", median_us="median", max_us="max")
        par = kt.pivot_table(index="kernel", columns="parity", values="us", aggfunc="mean").rename(columns={0: "even_us", 1: "odd_us"})
        piv = piv.join(sp).join(par)
    if out.endswith(".md"):   # machine-readable twin for bench.py's roofline.traffic
        import json
        json.dump({k: {c: float(v) for c, v in row.items() if v == v} for k, row in piv.to_dict(orient="index").items()}, open(out[:-3] + ".json", "w"), indent=1)
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc summary (mean per dispatch, aloam:: kernels only" + (f"; the last {last} dispatches of every kernel" if last > 0 else "") + ")\n\n```\n")
        f.write(piv.to_string(float_format=lambda v: f"{v:,.0f}"))
        f.write("\n```\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
