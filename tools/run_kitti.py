#!/usr/bin/env python
"""Run the MI355X path on a KITTI-odometry-style folder (the input side of reference src/kittiHelper.cpp, without ROS).

Layout read (reference src/kittiHelper.cpp:68-72,95-134, relative to --dataset):
    sequences/<seq>/times.txt                       one stamp per line
    velodyne/sequences/<seq>/velodyne/%06d.bin      float32 x, y, z, reflectance per point (16 B = the record the C ABI takes)
    results/<seq>.txt                               optional ground truth: 3x4 row-major camera-frame poses, one per line
Ground truth is moved into the lidar / "/camera_init" convention exactly like kittiHelper (R_transform = [0 0 1; -1 0 0; 0 -1 0],
:78-80,104-107).  Output: <out>/<seq>_odometry.txt and (with --mapping) <out>/<seq>_mapped.txt, one line `stamp tx ty tz qx qy qz qw`
per sweep, plus the ATE (RMSE of translation, same start, no alignment) against the ground truth when it is present.

KITTI is not part of this repository or image; `--selftest` writes a tiny synthetic sequence in this layout and runs on it.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R_TRANSFORM = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], float)


def read_times(path):
    """times.txt, one stamp per line, parsed like `stof(line)` (reference src/kittiHelper.cpp:88): single precision."""
    return [float(np.float32(l)) for l in open(path) if l.strip()]


def read_lidar(path):
    """One velodyne/%06d.bin: float32 x, y, z, reflectance per point (reference src/kittiHelper.cpp:25-35 read_lidar_data)."""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


def read_gt(path):
    """-> (N,3,3) rotations and (N,3) translations in the /camera_init convention of kittiHelper.cpp: every entry goes through
    `stof` (:95-102, single precision), then q = q_transform * q_w_i and t = q_transform * t (:104-107) with
    R_transform = [0 0 1; -1 0 0; 0 -1 0] (:78-80)."""
    P = np.loadtxt(path, dtype=np.float32, ndmin=2).astype(np.float64).reshape(-1, 3, 4)
    return np.einsum("ij,njk->nik", R_TRANSFORM, P[:, :, :3]), P[:, :, 3] @ R_TRANSFORM.T


def write_selftest(folder, seq="00", frames=6):
    syn = importlib.import_module("a-loam_amd.synthetic")
    scans, R, t, model = syn.make_sequence("HDL-64", frames, seed=77, columns=1024)
    os.makedirs(os.path.join(folder, "sequences", seq), exist_ok=True)
    os.makedirs(os.path.join(folder, "velodyne", "sequences", seq, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(folder, "results"), exist_ok=True)
    with open(os.path.join(folder, "sequences", seq, "times.txt"), "w") as f:
        f.writelines(f"{0.1 * k:e}\n" for k in range(frames))
    Rn, tn = R.numpy(), t.numpy()
    with open(os.path.join(folder, "results", seq + ".txt"), "w") as f:
        for k in range(frames):                       # camera-frame pose whose kittiHelper image is the lidar pose relative to frame 0
            Rl, tl = Rn[0].T @ Rn[k], Rn[0].T @ (tn[k] - tn[0])
            P = np.concatenate([R_TRANSFORM.T @ Rl, (R_TRANSFORM.T @ tl)[:, None]], 1)
            f.write(" ".join(f"{v:.9e}" for v in P.reshape(-1)) + "\n")
    for k, s in enumerate(scans):
        s.numpy().astype(np.float32).tofile(os.path.join(folder, "velodyne", "sequences", seq, "velodyne", f"{k:06d}.bin"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", help="dataset_folder of kitti_helper.launch")
    ap.add_argument("--seq", default="00")
    ap.add_argument("--out", default="kitti_out")
    ap.add_argument("--mapping", action="store_true")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--reference-order", action="store_true", help="sum voxel members in pcl::VoxelGrid's own order (the reference's bits; ~4x slower for one sensor): for runs that are compared pose by pose with A-LOAM's")
    ap.add_argument("--distortion", action="store_true", help="per-point interpolation ratio (the reference's DISTORTION 1; real KITTI sweeps are already de-skewed, so the reference ships 0)")
    args = ap.parse_args()
    if args.selftest:
        args.dataset = os.path.join(args.out, "selftest_dataset")
        write_selftest(args.dataset, args.seq)
    binding = importlib.import_module("a-loam_amd.binding")
    times = read_times(os.path.join(args.dataset, "sequences", args.seq, "times.txt"))
    if args.max_frames:
        times = times[: args.max_frames]
    gt_path = os.path.join(args.dataset, "results", args.seq + ".txt")
    gt = read_gt(gt_path) if os.path.exists(gt_path) else None
    # launch/aloam_velodyne_HDL_64.launch: scan_line 64, minimum_range 5, mapping resolutions 0.4 / 0.8
    gpu = binding.Aloam(n_scans=64, min_range=5.0, max_points=140000, distortion=args.distortion)
    if args.reference_order:
        gpu.set_voxel_sum_order(True)
    if args.mapping:
        gpu.mapping_enable(0.4, 0.8, pool_points=1 << 17)          # where the map starts: the pools double as it grows (src/laserMapping.cpp:737-783 push_back)
    os.makedirs(args.out, exist_ok=True)
    odo, mapped = [], []
    for k, stamp in enumerate(times):
        pts = read_lidar(os.path.join(args.dataset, "velodyne", "sequences", args.seq, "velodyne", f"{k:06d}.bin"))
        gpu.scan_register(pts)
        gpu.odometry_step()
        p = gpu.pose()
        odo.append([stamp, *p["t_w"], *p["q_w"]])
        if args.mapping:
            gpu.mapping_step()
            gpu.synchronize()
            m = gpu.map_pose()
            mapped.append([stamp, *m["t_w"], *m["q_w"]])
    np.savetxt(os.path.join(args.out, f"{args.seq}_odometry.txt"), np.array(odo), fmt="%.9e")
    if mapped:
        np.savetxt(os.path.join(args.out, f"{args.seq}_mapped.txt"), np.array(mapped), fmt="%.9e")
    if gt is not None:
        Rg, tg = gt
        tg = (tg[: len(odo)] - tg[0]) @ Rg[0]          # same start as the estimate (identity at the first sweep)
        for name, tr in (("odometry", odo), ("mapped", mapped)):
            if tr:
                e = np.array(tr)[:, 1:4] - tg
                print(f"{name}: {len(tr)} sweeps, ATE (RMSE, no alignment) = {np.sqrt((e ** 2).sum(1).mean()):.4f} m, final error = {np.linalg.norm(e[-1]):.4f} m")
    gpu.close()


if __name__ == "__main__":
    main()
