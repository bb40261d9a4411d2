"""Register / LDS / scratch figures of every kernel of a translation unit, from the code-object metadata (no GPU needed).
    python tools/kernel_regs.py odometry_kernels [substring] [-Dmacro=...]"""
import os, re, subprocess, sys, tempfile, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "--offload-device-only", "-S"]
tu = sys.argv[1]
sub = next((x for x in sys.argv[2:] if not x.startswith("-")), "")
extra = [x for x in sys.argv[2:] if x.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", out, os.path.join(ROOT, "a-loam_amd", "csrc", tu + ".hip")], check=True, capture_output=True)
    text = open(out).read()
    if "--keep" in sys.argv: open("/tmp/isa/" + tu + ".s", "w").write(text)
md = yaml.safe_load(re.search(r"\.amdgpu_metadata\n(.*?)\n\s*\.end_amdgpu_metadata", text, re.S).group(1))
names = [k[".name"] for k in md["amdhsa.kernels"]]
dem = subprocess.run(["c++filt", *names], capture_output=True, text=True).stdout.splitlines()
for dn, k in zip(dem, md["amdhsa.kernels"]):
    n = dn.split("(")[0].replace("void ", "").replace("aloam::", "")
    if sub in n:
        print(f"{n:48s} vgpr {k['.vgpr_count']:4d} agpr {k.get('.agpr_count', 0):3d} sgpr {k['.sgpr_count']:4d} spill {k['.vgpr_spill_count']:3d} scratch {k['.private_segment_fixed_size']:5d} lds {k['.group_segment_fixed_size']:6d}")
