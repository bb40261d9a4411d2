"""Per-dispatch durations of selected kernels from a rocprofv3 --kernel-trace --output-format csv run, in launch order (debug aid)."""
import csv, glob, sys
pat, sel = sys.argv[1], sys.argv[2:]
rows = []
for f in glob.glob(pat, recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if "aloam::" not in name:
                continue
            short = name.split("(")[0].replace("void aloam::", "")
            rows.append((int(r["Start_Timestamp"]), short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
# last full step: print the tail
tail = rows[-220:]
for t, n, d in tail:
    if not sel or any(s in n for s in sel):
        print(f"{n:45s} {d:9.1f} us")
