"""Regenerates tests/golden/ref_*.npz from oracle/_ref — outputs of the REFERENCE'S OWN translation units
(/root/reference/src/scanRegistration.cpp, src/laserOdometry.cpp + src/lidarFactor.hpp, compiled in place against the
stand-in third-party headers of oracle/ref_shim/) on small seeded synthetic sweeps.

Needs /root/reference (this container only); the resulting fixtures travel with the repo and pin both the oracle and
the HIP path wherever the tests run.
    python tools/make_ref_golden.py
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ref_py
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corr_index
syn = importlib.import_module("a-loam_amd.synthetic")

CASES = [("ref_vlp16_c600_seed11", "VLP-16", 4, 11, {"columns": 600}), ("ref_hdl64_c256_seed12", "HDL-64", 4, 12, {"columns": 256})]


def main():
    assert ref_py.build(), "oracle/_ref could not be built (is /root/reference present?)"
    for tag, name, frames, seed, kw in CASES:
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        xs = [s.numpy() for s in scans]
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range)
        odo = ref_py.laser_odometry(reg)
        out = {"R": R.numpy(), "t": t.numpy(), "n_scans": model.n_scans, "min_range": model.min_range, "frames": frames}
        for k in range(frames):
            out[f"scan{k}"] = xs[k]
            for key in ("sharp", "less_sharp", "flat", "less_flat", "curvature", "label"):
                out[f"{key}{k}"] = reg[k][key]
            out[f"cloud_intensity{k}"] = reg[k]["cloud"][:, 3].copy()      # xyz of the ring-ordered cloud are a permutation of the input
            out[f"cloud_xyz_sum{k}"] = reg[k]["cloud"][:, :3].astype(np.float64).sum(0)
            for key in ("q_lc", "t_lc", "q_w", "t_w"):
                out[f"{key}{k}"] = odo[k][key]
            out[f"corr{k}"] = np.array([odo[k]["corner_corr"], odo[k]["plane_corr"]])
            _store_indices(out, k, reg, odo)
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB")


def _store_indices(out, k, reg, odo):
    """closestPointInd / minPointInd2 / minPointInd3 of every factor of the frame's last ceres::Solve (laserOdometry.cpp:299-483), recovered from
    the constructor arguments inside the reference's residual blocks by exact coordinate look-up in the clouds they came from: rows
    (query index in the sharp / flat cloud of frame k, indices in laserCloudCornerLast / SurfLast = the less-sharp / less-flat cloud of frame k - 1)."""
    if k == 0:
        return
    e = corr_index.indices(odo[k]["edges"], reg[k]["sharp"], odo[k - 1]["corner_last"])
    p = corr_index.indices(odo[k]["planes"], reg[k]["flat"], odo[k - 1]["surf_last"])
    assert (e >= 0).all() and (p >= 0).all() and len(e) == odo[k]["corner_corr"] and len(p) == odo[k]["plane_corr"]
    out[f"edge_idx{k}"], out[f"plane_idx{k}"] = e.astype(np.int32), p.astype(np.int32)


DISTORT_CASES = [("refdistort_hdl64_c256_seed15", "HDL-64", 5, 15, {"columns": 256}), ("refdistort_vlp16_c600_seed16", "VLP-16", 5, 16, {"columns": 600})]


def main_distortion():
    """The de-skew branch of the reference's odometry node (src/laserOdometry.cpp:115-118,376-377,474-475), which ships compiled
    out by `#define DISTORTION 0` (:59): oracle/_ref/ref_laser_odometry_distort is the same translation unit with that one define
    flipped on its way into the compiler (oracle/Makefile).  Registration by the reference's scanRegistration.cpp as usual."""
    exe = os.path.join(ref_py.REF_DIR, "ref_laser_odometry_distort")
    for tag, name, frames, seed, kw in DISTORT_CASES:
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        xs = [s.numpy() for s in scans]
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range)
        odo = ref_py.laser_odometry(reg, exe=exe)
        plain = ref_py.laser_odometry(reg)
        out = {"R": R.numpy(), "t": t.numpy(), "n_scans": model.n_scans, "min_range": model.min_range, "frames": frames}
        for k in range(frames):
            out[f"scan{k}"] = xs[k]
            for key in ("q_lc", "t_lc", "q_w", "t_w"):
                out[f"{key}{k}"] = odo[k][key]
                out[f"plain_{key}{k}"] = plain[k][key]          # DISTORTION 0 on the same sweeps: shows the branch does something
            out[f"corr{k}"] = np.array([odo[k]["corner_corr"], odo[k]["plane_corr"]])
            _store_indices(out, k, reg, odo)
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB")


MAP_CASES = [("refmap_hdl64_c256_seed13", "HDL-64", 6, 13, {"columns": 256}, 0.4, 0.8), ("refmap_vlp16_c600_seed14", "VLP-16", 6, 14, {"columns": 600}, 0.2, 0.4)]


def main_mapping():
    """registration -> odometry -> mapping, all three by the reference's own translation units; the mapping inputs
    (what the node receives per frame) and outputs (pose, map<-odom transform, the whole cube map) are stored."""
    for tag, name, frames, seed, kw, line_res, plane_res in MAP_CASES:
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        xs = [s.numpy() for s in scans]
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range)
        odo = ref_py.laser_odometry(reg)
        fr = [dict(q_w=o["q_w"], t_w=o["t_w"], corner_last=o["corner_last"], surf_last=o["surf_last"], cloud=r["cloud"]) for o, r in zip(odo, reg)]
        mp = ref_py.laser_mapping(fr, line_res, plane_res)
        out = {"R": R.numpy(), "t": t.numpy(), "n_scans": model.n_scans, "min_range": model.min_range, "frames": frames, "line_res": line_res, "plane_res": plane_res}
        for k in range(frames):
            out[f"odom_q{k}"], out[f"odom_t{k}"] = fr[k]["q_w"], fr[k]["t_w"]
            out[f"corner_last{k}"], out[f"surf_last{k}"], out[f"full{k}"] = fr[k]["corner_last"], fr[k]["surf_last"], fr[k]["cloud"]
            for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
                out[f"{key}{k}"] = mp[k][key]
            out[f"cen{k}"] = np.array(mp[k]["cen"])
            out[f"registered_s7_{k}"] = mp[k]["registered"][::7].copy()
            for nm in ("corner_map", "surf_map"):
                ids = sorted(mp[k][nm])
                out[f"{nm}_ids{k}"] = np.array(ids, np.int32)
                out[f"{nm}_cnt{k}"] = np.array([len(mp[k][nm][c]) for c in ids], np.int32)
                out[f"{nm}_pts{k}"] = np.concatenate([mp[k][nm][c] for c in ids]) if ids else np.zeros((0, 4), np.float32)
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB")


def main_factors():
    """The reference's LidarEdgeFactor / LidarPlaneFactor with a general interpolation ratio s (the branch its nodes compile out
    with #define DISTORTION 0): residuals and per-block Jacobians from its own templates -> tests/golden/reffactor_s.npz."""
    rng = np.random.default_rng(2024)
    n = 400
    rec = np.zeros((n, 21))
    rec[:, 0] = np.arange(n) % 2
    rec[:, 1] = np.concatenate([rng.uniform(0.0, 1.0, n - 6), [0.0, 1.0, 0.5, 1e-9, 1.0 - 1e-12, 0.25]])
    ang = rng.uniform(0, 0.2, n); ang[:8] = [0.0, 1e-9, 1e-8, 3e-8, 1e-7, 1e-4, 3.0, 3.14]     # incl. the (1 - eps) guard of slerp and q.w < 0
    ax = rng.normal(size=(n, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    rec[:, 2:5] = ax * np.sin(ang / 2)[:, None]; rec[:, 5] = np.cos(ang / 2)
    rec[8:16, 2:6] *= -1.0                                                                   # d < 0 branch
    rec[:, 6:9] = rng.normal(scale=0.5, size=(n, 3))
    rec[:, 9:21] = rng.uniform(-30, 30, (n, 12))
    rec[:, 12:21] = rec[:, 9:12].repeat(3, 0).reshape(n, 9) + rng.normal(scale=1.0, size=(n, 9))   # neighbours near the point
    r, jq, jt = ref_py.lidar_factors(rec)
    path = os.path.join(ROOT, "tests", "golden", "reffactor_s.npz")
    np.savez_compressed(path, records=rec, residual=r, jac_q=jq, jac_t=jt)
    print(path, os.path.getsize(path) // 1024, "KiB")


# Sweeps of the BENCHMARKED size (BASELINE.json configs[1]: 64 x 2048) through the reference's own code, plus the three shapes the small
# fixtures above do not hold: KITTI-shaped irregular sweeps at full size, the 32-line ring formula (src/scanRegistration.cpp:175-185), and
# a sweep that starts just beyond the +-pi wrap of atan2 (:141-153,208-236).  The raw sweeps are NOT stored: a test regenerates them from
# the seed (a-loam_amd/synthetic.py is deterministic on the CPU) and checks the sha256 stored here; of the big less-flat clouds only
# frame 1 is stored (the <= 4 ulp comparison needs values), the other frames by hash of the reference's bits + size + integer intensities.
FULL_CASES = [("reffull_hdl64_seed31", "HDL-64", 3, 31, {}),
              ("reffull_hdl64_rough_seed32", "HDL-64", 3, 32, {"rough": True}),
              ("reffull_hdl32_c1024_seed33", "HDL-32", 3, 33, {"columns": 1024}),
              ("reffull_hdl64_wrap_seed34", "HDL-64", 3, 34, {"az_offset": 0.75})]


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main_full():
    import json
    assert ref_py.build(), "oracle/_ref could not be built (is /root/reference present?)"
    for tag, name, frames, seed, kw in FULL_CASES:
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        xs = [s.numpy() for s in scans]
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range)
        odo = ref_py.laser_odometry(reg)
        out = {"sensor": name, "seed": seed, "kwargs": json.dumps(kw), "n_scans": model.n_scans, "min_range": model.min_range, "frames": frames,
               "max_points": max(len(x) for x in xs)}
        for k in range(frames):
            out[f"scan_sha{k}"], out[f"scan_n{k}"] = sha(xs[k]), len(xs[k])
            for key in ("sharp", "less_sharp", "flat"):
                out[f"{key}{k}"] = reg[k][key]
            lf = reg[k]["less_flat"]
            out[f"less_flat_sha{k}"], out[f"less_flat_n{k}"] = sha(lf), len(lf)
            out[f"less_flat_int{k}"] = sha(lf[:, 3].astype(np.int32))
            if k == 1:
                out[f"less_flat{k}"] = lf
            n = len(reg[k]["curvature"])
            out[f"cloud_n{k}"] = len(reg[k]["cloud"])
            out[f"cloud_sha{k}"] = sha(reg[k]["cloud"])
            out[f"curvature_sha{k}"], out[f"label_sha{k}"] = sha(reg[k]["curvature"][:n]), sha(reg[k]["label"][:n].astype(np.int32))
            for key in ("q_lc", "t_lc", "q_w", "t_w"):
                out[f"{key}{k}"] = odo[k][key]
            out[f"corr{k}"] = np.array([odo[k]["corner_corr"], odo[k]["plane_corr"]])
            out[f"corner_last_sha{k}"], out[f"surf_last_sha{k}"] = sha(odo[k]["corner_last"]), sha(odo[k]["surf_last"])
            _store_indices(out, k, reg, odo)
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB", [int(out[f"scan_n{k}"]) for k in range(frames)], [list(out[f"corr{k}"]) for k in range(frames)])


# The whole chain registration -> odometry -> mapping by the reference's three translation units on sweeps of the benchmarked size: refined poses,
# map<-odom transforms, cube window and the cube map (ids, populations, sha256 of every class's points in cube order) after every frame.
FULL_MAP_CASES = [("reffullmap_hdl64_seed41", "HDL-64", 4, 41, {}, 0.4, 0.8)]


def main_full_mapping():
    import json
    assert ref_py.build()
    for tag, name, frames, seed, kw, line_res, plane_res in FULL_MAP_CASES:
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        xs = [s.numpy() for s in scans]
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range)
        odo = ref_py.laser_odometry(reg)
        fr = [dict(q_w=o["q_w"], t_w=o["t_w"], corner_last=o["corner_last"], surf_last=o["surf_last"], cloud=r["cloud"]) for o, r in zip(odo, reg)]
        mp = ref_py.laser_mapping(fr, line_res, plane_res)
        out = {"sensor": name, "seed": seed, "kwargs": json.dumps(kw), "n_scans": model.n_scans, "min_range": model.min_range, "frames": frames,
               "max_points": max(len(x) for x in xs), "line_res": line_res, "plane_res": plane_res}
        for k in range(frames):
            out[f"scan_sha{k}"], out[f"scan_n{k}"] = sha(xs[k]), len(xs[k])
            out[f"odom_q{k}"], out[f"odom_t{k}"] = fr[k]["q_w"], fr[k]["t_w"]
            for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
                out[f"{key}{k}"] = mp[k][key]
            out[f"cen{k}"] = np.array(mp[k]["cen"])
            out[f"registered_sha{k}"], out[f"registered_n{k}"] = sha(mp[k]["registered"]), len(mp[k]["registered"])
            for nm in ("corner_map", "surf_map"):
                ids = sorted(mp[k][nm])
                out[f"{nm}_ids{k}"] = np.array(ids, np.int32)
                out[f"{nm}_cnt{k}"] = np.array([len(mp[k][nm][c]) for c in ids], np.int32)
                out[f"{nm}_sha{k}"] = sha(np.concatenate([mp[k][nm][c] for c in ids]) if ids else np.zeros((0, 4), np.float32))
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB", [int(out[f"corner_map_cnt{k}"].sum()) for k in range(frames)], [int(out[f"surf_map_cnt{k}"].sum()) for k in range(frames)])


# Long horizon on a TRAVELLING sensor: the reference's three translation units end to end over hundreds of sweeps of a non-returning drive
# (a-loam_amd/synthetic.py `travel`: ~445 m down a 700 m street), so that the pose chain integrates without renormalisation for 300 frames
# (src/laserOdometry.cpp:504-505), the submap is gathered from changing cubes (src/laserMapping.cpp:509-539), the cubes grow and are re-filtered
# frame after frame (:737-801) and the cube window shifts because the sensor really is 375 m from where it started (:323-507).
# Stored: every frame's odometry and refined pose, map<-odom transform, window centre and class totals; every CHECK frames (and the last) the cube
# ids / populations and the sha256 of each class's points in cube order, the registered cloud's, and the sweep's own sha256.
LONG_CASES = [("reflong_hdl64_c512_seed51", "HDL-64", 300, 51, {"columns": 512, "travel": True, "step": 1.6}, 0.4, 0.8, 25)]


def main_long():
    import json, time
    assert ref_py.build()
    for tag, name, frames, seed, kw, line_res, plane_res, check in LONG_CASES:
        t0 = time.time()
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        xs = [s.numpy() for s in scans]
        t1 = time.time()
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range)
        odo = ref_py.laser_odometry(reg)
        t2 = time.time()
        fr = [dict(q_w=o["q_w"], t_w=o["t_w"], corner_last=o["corner_last"], surf_last=o["surf_last"], cloud=r["cloud"]) for o, r in zip(odo, reg)]
        mp = ref_py.laser_mapping(fr, line_res, plane_res)
        t3 = time.time()
        # The reference's own sensitivity, as the yardstick for any free-running comparison over this horizon: the same three translation units
        # on the same sweeps with ONE coordinate of one point in a hundred moved by ONE ulp (a different last bit of the sensor driver's float
        # conversion).  Feature selection and every threshold of the pipeline are discrete decisions on those bits, so the two runs of the
        # reference part ways at once and differ by centimetres after 300 frames.
        ys = [x.copy() for x in xs]
        for k in range(frames):
            ys[k][k % 100::100, 0] = np.nextafter(ys[k][k % 100::100, 0], np.float32(1e9))
        reg_u = ref_py.scan_registration(ys, model.n_scans, model.min_range)
        odo_u = ref_py.laser_odometry(reg_u)
        mp_u = ref_py.laser_mapping([dict(q_w=o["q_w"], t_w=o["t_w"], corner_last=o["corner_last"], surf_last=o["surf_last"], cloud=r["cloud"])
                                     for o, r in zip(odo_u, reg_u)], line_res, plane_res, dump_map=False)
        out = {"sensor": name, "seed": seed, "kwargs": json.dumps(kw), "n_scans": model.n_scans, "min_range": model.min_range, "frames": frames,
               "max_points": max(len(x) for x in xs), "line_res": line_res, "plane_res": plane_res, "check": check,
               "ulp_t_w": np.stack([m["t_w"] for m in mp_u]), "ulp_q_w": np.stack([m["q_w"] for m in mp_u]),
               "ulp_odom_t": np.stack([o["t_w"] for o in odo_u]), "ulp_odom_q": np.stack([o["q_w"] for o in odo_u]),
               "gt_R": R.numpy(), "gt_t": t.numpy(),
               "odom_q": np.stack([f["q_w"] for f in fr]), "odom_t": np.stack([f["t_w"] for f in fr]),
               "q_w": np.stack([m["q_w"] for m in mp]), "t_w": np.stack([m["t_w"] for m in mp]),
               "q_wmap_wodom": np.stack([m["q_wmap_wodom"] for m in mp]), "t_wmap_wodom": np.stack([m["t_wmap_wodom"] for m in mp]),
               "cen": np.array([m["cen"] for m in mp], np.int32),
               "corr": np.array([[o["corner_corr"], o["plane_corr"]] for o in odo], np.int32),
               "scan_n": np.array([len(x) for x in xs], np.int32),
               "map_total": np.array([[sum(len(v) for v in m["corner_map"].values()), sum(len(v) for v in m["surf_map"].values())] for m in mp], np.int32),
               "map_cubes": np.array([[len(m["corner_map"]), len(m["surf_map"])] for m in mp], np.int32)}
        for k in sorted(set(range(0, frames, check)) | {frames - 1}):
            out[f"scan_sha{k}"] = sha(xs[k])
            out[f"registered_sha{k}"], out[f"registered_n{k}"] = sha(mp[k]["registered"]), len(mp[k]["registered"])
            for nm in ("corner_map", "surf_map"):
                ids = sorted(mp[k][nm])
                out[f"{nm}_ids{k}"] = np.array(ids, np.int32)
                out[f"{nm}_cnt{k}"] = np.array([len(mp[k][nm][c]) for c in ids], np.int32)
                out[f"{nm}_sha{k}"] = sha(np.concatenate([mp[k][nm][c] for c in ids]) if ids else np.zeros((0, 4), np.float32))
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        err = np.linalg.norm(out["t_w"] - (R[0].numpy().T @ (t.numpy() - t[0].numpy()).T).T, axis=1)
        print(path, os.path.getsize(path) // 1024, "KiB; render %.0f s, registration + odometry %.0f s, mapping %.0f s" % (t1 - t0, t2 - t1, t3 - t2))
        du = np.linalg.norm(out["ulp_t_w"] - out["t_w"], axis=1)
        print("  the reference against itself with 1 %% of the input points moved by one ulp: refined pose differs by %.2e m at frame 25, %.2e at 100, max %.2e; first frame beyond 1e-4 m: %d"
              % (du[25], du[100], du.max(), int(np.argmax(du > 1e-4))))
        print("  window centre first / last", out["cen"][0], out["cen"][-1], "first shift at frame", int(np.argmax((out["cen"] != out["cen"][0]).any(1))),
              "; map totals last", out["map_total"][-1], "; refined-pose error vs ground truth: median %.3f m, last %.3f m" % (np.median(err), err[-1]))


if __name__ == "__main__":
    if "--long" in sys.argv:
        main_long()
    elif "--full-mapping" in sys.argv:
        main_full_mapping()
    elif "--full" in sys.argv:
        main_full()
        main_full_mapping()
    else:
        main()
        main_mapping()
        main_factors()
        main_distortion()
        main_full()
        main_full_mapping()
        main_long()
