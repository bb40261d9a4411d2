#!/usr/bin/env python
"""A/B of two builds of the library on the GPU box WITHOUT torch and pytest: same sweeps through both, every output compared bit
for bit, per-kernel hipEvent times side by side.  A call costs ~20-30 s of box time (python + numpy + ctypes start in a second;
`import torch` alone takes a minute on a fresh box), so a kernel experiment can be answered ten times as often as with
tools/gpu_variants.sh; the full parity suite still decides what ships.

    # here (CPU, torch available): sweeps of a few distinct synthetic sequences -> tools/_ab_inputs.npz (git-ignored, travels with gpurun)
    python tools/ab_check.py make-inputs [--sensor HDL-64] [--frames 3] [--sequences 4]
    # on the GPU box, e.g.  gpurun --timeout 60 -- 'python tools/ab_check.py ab product a-loam_amd/lib/variants/libX.so'
    python tools/ab_check.py run <lib | product> <out.npz> [--batch 1024] [--steps 12] [--mapping] [--cube-hist]   # --cube-hist: sizes of the map cubes at the end
    # ALOAM_AB_WATCHDOG=<seconds>: dump the Python stacks and exit if a run has not finished by then (the first process on a cold box can take a minute to page the HIP runtime in)
    python tools/ab_check.py compare <a.npz> <b.npz>
    python tools/ab_check.py ab <lib a> <lib b> [run options]        # run + run + compare, each run in its own process

The batch is filled by replicating the stored sequences over device memory (distinct addresses, so caches see a real batch; sequence
b replays stored sequence b mod S).  What is compared: the four feature clouds and CORNER_LAST / SURF_LAST of three sequences after
every step, poses and solver statistics of those sequences after every step, (with --mapping) the refined poses and map sizes.
"""
from __future__ import annotations

import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
INPUTS = os.environ.get("ALOAM_AB_INPUTS", os.path.join(ROOT, "tools", "_ab_inputs.npz"))   # ALOAM_AB_INPUTS: another input set, e.g. the travelling drive


def opt(argv, name, default, cast=int):
    return cast(argv[argv.index(name) + 1]) if name in argv else default


def make_inputs(argv):
    syn = importlib.import_module("a-loam_amd.synthetic")
    sensor, T, S = opt(argv, "--sensor", "HDL-64", str), opt(argv, "--frames", 3), opt(argv, "--sequences", 4)
    out = {}
    travel = "--travel" in argv           # sweeps of the travelling drive (a-loam_amd/synthetic.py `travel`), skipping the slow first `--skip` frames
    skip = opt(argv, "--skip", 20)
    for s in range(S):
        if travel:
            scans, _, _, model = syn.make_sequence(sensor, skip + T, seed=100 + s, travel=True, step=1.6)
            scans = scans[skip:]
        else:
            scans, _, _, model = syn.make_sequence(sensor, T, seed=100 + s)
        for k, x in enumerate(scans):
            out[f"s{s}_f{k}"] = x.numpy().astype(np.float32)
    out["meta"] = np.array(json.dumps({"sensor": sensor, "frames": T, "sequences": S, "n_scans": model.n_scans, "min_range": model.min_range,
                                       "ring_from_field": bool(model.ring_from_field), "columns": model.columns}))
    np.savez_compressed(INPUTS, **out)
    print(INPUTS, os.path.getsize(INPUTS) >> 20, "MiB")


class Hip:
    """the three runtime calls the harness needs, through ctypes"""

    def __init__(self):
        self.l = C.CDLL("libamdhip64.so")
        self.l.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.l.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.l.hipFree.argtypes = [C.c_void_p]

    def malloc(self, n):
        p = C.c_void_p()
        rc = self.l.hipMalloc(C.byref(p), n)
        assert rc == 0 and p.value, f"hipMalloc({n}) -> {rc}"
        return p.value

    def h2d(self, dst, arr):
        assert self.l.hipMemcpy(dst, arr.ctypes.data, arr.nbytes, 1) == 0

    def d2d(self, dst, src, n):
        assert self.l.hipMemcpy(dst, src, n, 3) == 0

    def free(self, p):
        self.l.hipFree(p)


def run(argv):
    lib_path, out_path = argv[0], argv[1]
    if lib_path != "product":
        os.environ["ALOAM_MI355X_LIB"] = os.path.abspath(lib_path)
    B, steps, mapping = opt(argv, "--batch", 1024), opt(argv, "--steps", 12), "--mapping" in argv
    binding = importlib.import_module("a-loam_amd.binding")
    z = np.load(INPUTS)
    meta = json.loads(str(z["meta"]))
    T, S = meta["frames"], meta["sequences"]
    NP = max(z[f"s{s}_f{k}"].shape[0] for s in range(S) for k in range(T))
    NP = (NP + 255) // 256 * 256 + int(os.environ.get("ALOAM_AB_INPAD", "0"))   # ALOAM_AB_INPAD: the stored sweeps off the power-of-two stride too
    hip = Hip()
    seq_stride = T * NP * 16
    base = hip.malloc(B * seq_stride)
    counts = np.zeros((B, T), np.int32)
    for s in range(min(S, B)):                                   # stored sequence s -> batch slot s, then device-to-device into s + S, s + 2 S, ...
        for k in range(T):
            x = np.ascontiguousarray(z[f"s{s}_f{k}"][:, :4], np.float32)
            hip.h2d(base + s * seq_stride + k * NP * 16, x)
            counts[s::S, k] = x.shape[0]
        for b in range(s + S, B, S):
            hip.d2d(base + b * seq_stride, base + s * seq_stride, seq_stride)
    gpu = binding.Aloam(n_scans=meta["n_scans"], min_range=meta["min_range"], ring_from_field=meta["ring_from_field"], batch=B, max_points=NP - int(os.environ.get("ALOAM_AB_INPAD", "0")) + int(os.environ.get("ALOAM_AB_PAD", "0")),   # ALOAM_AB_PAD: per-sequence strides off the power of two
                        max_ring_points=2059 if meta["columns"] <= 2048 else 4107)
    if mapping:
        gpu.mapping_enable(0.4, 0.8, pool_points=262144 + int(os.environ.get("ALOAM_AB_POOLPAD", "0")))
    order, t, d = [], 0, 1                                       # ping-pong replay of the stored frames, like bench.py's frame_order
    for _ in range(steps):
        order.append(t)
        if T > 1:
            if t + d < 0 or t + d >= T:
                d = -d
            t += d
    watch = sorted({0, min(1, B - 1), B - 1})
    rec = {}
    nin = {k: (C.c_int * B)(*[int(v) for v in counts[:, k]]) for k in range(T)}
    gpu.profile_enable(True)
    t0 = time.perf_counter()
    for i, k in enumerate(order):
        gpu.process_device(base + k * NP * 16, seq_stride, nin[k])
        if mapping:
            gpu.mapping_step()
        if i < 2 * T or i == steps - 1:                          # outputs of the first two passes over the frames and of the last step
            gpu.synchronize()
            for b in watch:
                f = gpu.features(b)
                for name in ("sharp", "less_sharp", "flat", "less_flat"):
                    rec[f"step{i}_seq{b}_{name}"] = f[name]
                rec[f"step{i}_seq{b}_corner_last"] = gpu.cloud(binding.CLOUD_CORNER_LAST, b)
                rec[f"step{i}_seq{b}_surf_last"] = gpu.cloud(binding.CLOUD_SURF_LAST, b)
                p = gpu.pose(b)
                rec[f"step{i}_seq{b}_pose"] = np.concatenate([p["q_w"], p["t_w"], p["q_lc"], p["t_lc"]])
                st = gpu.odom_stats(b)
                rec[f"step{i}_seq{b}_stats"] = np.array([v for key in sorted(st) for v in st[key]], np.float64)
                if mapping:
                    m = gpu.map_pose(b)
                    rec[f"step{i}_seq{b}_map_pose"] = np.concatenate([m["q_w"], m["t_w"]])
                    rec[f"step{i}_seq{b}_map_info"] = np.array(list(gpu.map_info(b).values()), np.int64)
    gpu.synchronize()
    wall = time.perf_counter() - t0
    prof = gpu.profile()
    rec["profile"] = np.array(json.dumps({"lib": lib_path, "batch": B, "steps": steps, "mapping": mapping, "wall_s_with_readbacks": wall,
                                          "ms_per_step": {k: v["total_ms"] / steps for k, v in prof.items() if v["launches"]}}))
    np.savez(out_path, **rec)
    L = binding.lib()
    if hasattr(L, "aloam_debug_assoc_stats"):                   # -DALOAM_ASSOC_STATS builds: what the association waves did
        st = (C.c_ulonglong * 64)()
        L.aloam_debug_assoc_stats(st)
        names = {0: "pairs", 1: "queries", 2: "has1", 3: "kept_ok", 4: "fine sweeps", 5: "fine buckets", 6: "fine candidates", 7: "fine rows32", 20: "ring calls", 21: "ring want2",
                 22: "ring want3", 23: "ring had bound", 24: "stage0", 25: "stage1", 26: "stage2", 28: "coarse calls",
                 8: "s0 sweeps", 9: "s0 buckets", 10: "s0 cand", 11: "s0 rows64", 12: "s1 sweeps", 13: "s1 buckets", 14: "s1 cand", 15: "s1 rows64", 16: "s2 sweeps", 17: "s2 buckets", 18: "s2 cand", 19: "s2 rows64"}
        for cls in (0, 1):
            q = max(1, st[cls * 32 + 1])
            print("corner" if cls == 0 else "plane", {names[i]: round(st[cls * 32 + i] / q, 3) for i in sorted(names)}, "queries", st[cls * 32 + 1])
    if hasattr(L, "aloam_debug_phase_clock"):                   # -DALOAM_PHASE_CLOCK builds: mean shader-clock time between the phase markers of k_ring_features
        st = (C.c_ulonglong * 64)()
        L.aloam_debug_phase_clock(st)
        names = ["ticket_taken", "after_curvature", "after_reach", "after_select1", "after_redo_labels_counts", "after_bbox_voxidx", "after_run_heads", "after_sort",
                 "after_vox_heads_gather", "after_centroids", "picks_out"]

        def slot(s):
            h = 0
            for c in s.encode():
                h = (h * 33 + c) & 0xffffffff
            return h % 32
        assert len({slot(n) for n in names}) == len(names)
        cnt = [st[32 + slot(n)] for n in names]
        print("phase clock: workgroups per marker", cnt)
        tot = 0.0
        for a_, b_ in zip(names[:-1], names[1:]):
            if st[32 + slot(a_)] == st[32 + slot(b_)] and st[32 + slot(a_)]:
                d = ((st[slot(b_)] - st[slot(a_)]) & (2 ** 64 - 1)) / st[32 + slot(a_)]
                tot += d
                print(f"  {a_:28s} -> {b_:28s} {d:12.0f} clocks")
        print(f"  total {tot:.0f} clocks per workgroup")
    if hasattr(L, "aloam_debug_phase_clock_odo"):               # the same for k_build_grids_fused (surf class): numbered points
        st = (C.c_ulonglong * 64)()
        L.aloam_debug_phase_clock_odo(st)
        names = ["entry", "tables zeroed", "counted", "flags / walk tables", "scanned", "filled (thread 0)", "filled (all)"]
        for i in range(1, len(names)):
            if st[32 + i] and st[32 + i] == st[32 + i - 1]:
                print(f"  grid build  {names[i - 1]:22s} -> {names[i]:22s} {((st[i] - st[i - 1]) & (2 ** 64 - 1)) / st[32 + i]:12.0f} clocks")
    if mapping and "--cube-hist" in argv:                        # sizes of the map cubes the per-cube re-filter works on (k_vox_lds instances: <= 2048 / <= 8192 / <= 65536 points)
        L.aloam_map_cube_counts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        for b in watch:
            for cls in (0, 1):
                cnt = np.zeros(21 * 21 * 11, np.int32)
                L.aloam_map_cube_counts(gpu.h, b, cls, cnt.ctypes.data)
                nz = np.sort(cnt[cnt > 0])[::-1]
                print(f"seq {b} {'corner' if cls == 0 else 'surf'}: {len(nz)} cubes, {int(nz.sum())} points; sizes", nz.tolist())
    gpu.close()
    hip.free(base)
    print(lib_path, "->", out_path, "kernel ms per step:", round(sum(v["total_ms"] for v in prof.values()) / steps, 3))


def compare(argv):
    a, b = np.load(argv[0]), np.load(argv[1])
    pa, pb = json.loads(str(a["profile"])), json.loads(str(b["profile"]))
    keys = sorted(k for k in a.files if k != "profile")
    assert keys == sorted(k for k in b.files if k != "profile"), "the two runs recorded different outputs"
    bad = [k for k in keys if a[k].shape != b[k].shape or not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))]
    print(f"{len(keys) - len(bad)} of {len(keys)} recorded arrays bit-identical" + (f"; DIFFERENT: {bad[:12]}" if bad else ""))
    print(f"{'kernel':28s} {pa['lib'][-28:]:>28s} {pb['lib'][-28:]:>28s}   (ms per {pa['batch']}-sequence step)")
    for k in pa["ms_per_step"]:
        x, y = pa["ms_per_step"][k], pb["ms_per_step"].get(k, float("nan"))
        print(f"{k:28s} {x:28.4f} {y:28.4f}   {100 * (y - x) / x if x else 0:+.1f} %")
    sa, sb = sum(pa["ms_per_step"].values()), sum(pb["ms_per_step"].values())
    print(f"{'sum':28s} {sa:28.4f} {sb:28.4f}   {100 * (sb - sa) / sa:+.1f} %")
    return 1 if bad else 0


def ab(argv):
    libs, rest = argv[:2], argv[2:]
    outs = []
    for i, l in enumerate(libs):
        out = os.path.join(ROOT, "gpurun_out", f"ab_{i}.npz")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run", l, out, *rest])
        if r.returncode:
            raise SystemExit(f"run with {l} failed")
        outs.append(out)
    return compare(outs)


if __name__ == "__main__":
    if os.environ.get("ALOAM_AB_WATCHDOG"):                      # seconds: dump every thread's Python stack and exit if the run has not finished by then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["ALOAM_AB_WATCHDOG"]), exit=True)
    cmd, args = sys.argv[1], sys.argv[2:]
    sys.exit({"make-inputs": make_inputs, "run": run, "compare": compare, "ab": ab}[cmd](args) or 0)
