"""Static instruction accounting of one kernel of a translation unit: VALU / SALU / LDS / VMEM instruction counts between
consecutive s_barrier instructions of the gfx950 code hipcc generates (no GPU needed).  This is the view the k_ring_features
instruction-count pass of round 3 worked from: a phase whose count is out of proportion to what it computes —
fully unrolled divergent LDS walks, wave-uniform values carried in VGPRs, selects over a whole register tuple — stands out at once.
    python tools/isa_segments.py registration_kernels 'k_ring_features<2048>' [--dump N]     # --dump N: print segment N
Counts are static (a loop body counts once); multiply by trip counts by hand."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "--offload-device-only", "-S"]


def kernel_text(tu, kernel):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, os.path.join(ROOT, "a-loam_amd", "csrc", tu + ".hip")], check=True, capture_output=True)
        lines = open(out).read().split("\n")
    labels = [(i, l[:-1].split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt", *[n for _, n in labels]], capture_output=True, text=True).stdout.splitlines()
    for (i, mangled), dem in zip(labels, names):
        if kernel in dem:
            end = next(j for j in range(i, len(lines)) if lines[j].startswith("\t.set " + mangled + "."))
            return lines[i:end]
    raise SystemExit(f"{kernel} not found in {tu}; kernels: {names}")


def main():
    tu, kernel = sys.argv[1], sys.argv[2]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else None
    lines = kernel_text(tu, kernel)
    segs, cur = [], {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "first": 0}
    for i, l in enumerate(lines):
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            cur["last"] = i; segs.append(cur); cur = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "first": i}
        elif op.startswith("v_"): cur["valu"] += 1
        elif op.startswith("s_") and op not in ("s_nop", "s_waitcnt"): cur["salu"] += 1
        elif op.startswith("ds_"): cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vmem"] += 1
    cur["last"] = len(lines); segs.append(cur)
    print(f"{kernel}: {len(lines)} lines, {len(segs)} segments between barriers")
    for k, s in enumerate(segs):
        print(f"  {k:3d}  lines {s['first']:6d}-{s['last']:6d}  VALU {s['valu']:5d}  SALU {s['salu']:5d}  LDS {s['lds']:4d}  VMEM {s['vmem']:4d}")
    if dump is not None:
        print("\n".join(lines[segs[dump]["first"]:segs[dump]["last"] + 1]))


if __name__ == "__main__":
    main()
