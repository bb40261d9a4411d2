"""VALU / SALU / LDS / VMEM instruction counts of a kernel between the `; ##PHASE name` markers its source leaves in the generated code
(ALOAM_PHASE in the kernels; static counts in layout order: a loop body counts once, a block the compiler moved behind a later marker is
counted there).    python tools/isa_phases.py odometry_kernels 'k_associate_pair<true, false>' [-Dmacro=..]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "--offload-device-only", "-S"]
tu, kernel = sys.argv[1], sys.argv[2]
extra = [x for x in sys.argv[3:] if x.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", out, os.path.join(ROOT, "a-loam_amd", "csrc", tu + ".hip")], check=True, capture_output=True)
    lines = open(out).read().split("\n")
labels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
names = subprocess.run(["c++filt", *[n for _, n in labels]], capture_output=True, text=True).stdout.splitlines()
start = next(i for (i, _), dn in zip(labels, names) if kernel in dn)
end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
cur, order, acc = "entry", ["entry"], {"entry": [0, 0, 0, 0]}
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; ##PHASE (\S+)", t)
    if m:
        cur = m.group(1)
        if cur not in acc: acc[cur] = [0, 0, 0, 0]; order.append(cur)
        continue
    if not t or t[0] in ";." or t.endswith(":"): continue
    op = t.split()[0]
    k = 0 if op.startswith("v_") else 1 if (op.startswith("s_") and op not in ("s_nop", "s_waitcnt")) else 2 if op.startswith("ds_") else 3 if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else None
    if k is not None: acc[cur][k] += 1
print(f"{kernel}: static counts between phase markers")
for n in order: print(f"  {n:12s} VALU {acc[n][0]:5d}  SALU {acc[n][1]:5d}  LDS {acc[n][2]:4d}  VMEM {acc[n][3]:4d}")
