#!/usr/bin/env python
"""Run the MI355X path on the `/velodyne_points` messages of a ROS 1 bag (what `rosbag play` feeds the reference's nodes,
reference README.md:36-47), without ROS; and write such a bag from a KITTI-odometry folder (the `to_bag` branch of reference
src/kittiHelper.cpp:74-76,164-176, lidar topic only).

    python tools/run_bag.py --bag nsh_indoor_outdoor.bag --scan-line 16 --minimum-range 0.3 [--mapping] [--out bag_out]
    python tools/run_bag.py --from-kitti <dataset_folder> --seq 00 --write-bag kitti_00.bag
    python tools/run_bag.py --selftest            # writes a small synthetic VLP-16 bag, reads it back and runs on it (needs the GPU)

Every message is decoded like pcl::fromROSMsg into pcl::PointXYZ does (x, y, z by field name, reference src/scanRegistration.cpp:132-133)
and handed to the C ABI as 12-byte records; output: <out>/odometry.txt (and mapped.txt with --mapping), one line
`stamp tx ty tz qx qy qz qw` per sweep, stamp = header.stamp of the cloud (src/scanRegistration.cpp:415, src/laserOdometry.cpp:514).
Defaults are launch/aloam_velodyne_VLP_16.launch (scan_line 16, minimum_range 0.3, mapping resolutions 0.2 / 0.4).
The bag code (a-loam_amd/rosbag1.py) follows the published format; this image has no ROS to cross-check it against.
"""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rosbag1 = importlib.import_module("a-loam_amd.rosbag1")


def sweeps(bag_path, topic):
    """(header stamp in s, (N, 3) float32 x y z) for every PointCloud2 on `topic`, in recording order."""
    for _, typ, _, raw in rosbag1.read_messages(bag_path, [topic]):
        if typ and typ != rosbag1.POINTCLOUD2_TYPE:
            raise rosbag1.BagError(f"{topic} carries {typ}, not {rosbag1.POINTCLOUD2_TYPE}")
        msg = rosbag1.decode_pointcloud2(raw)
        yield msg["stamp_ns"] * 1e-9, rosbag1.pointcloud2_xyz(msg)


def write_bag(path, clouds, stamps, topic="/velodyne_points", compression="none"):
    """clouds: (N, 4) float32 x y z intensity each; one sensor_msgs/PointCloud2 in pcl::toROSMsg<PointXYZI> layout per sweep,
    header.stamp = bag time = the sweep's stamp, frame /camera_init (src/kittiHelper.cpp:152-156,168)."""
    with rosbag1.BagWriter(path, compression=compression) as w:
        for k, (pts, t) in enumerate(zip(clouds, stamps)):
            t_ns = int(round(float(t) * 1e9))
            w.write(topic, rosbag1.POINTCLOUD2_TYPE, rosbag1.POINTCLOUD2_MD5, rosbag1.POINTCLOUD2_DEFINITION, t_ns,
                    rosbag1.encode_pointcloud2_xyzi(pts, t_ns, "/camera_init", k))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bag")
    ap.add_argument("--topic", default="/velodyne_points")
    ap.add_argument("--scan-line", type=int, default=16)
    ap.add_argument("--minimum-range", type=float, default=0.3)
    ap.add_argument("--line-res", type=float, default=0.2)
    ap.add_argument("--plane-res", type=float, default=0.4)
    ap.add_argument("--mapping", action="store_true")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--out", default="bag_out")
    ap.add_argument("--from-kitti", help="dataset_folder of kitti_helper.launch: write its lidar sweeps as a bag instead of running")
    ap.add_argument("--seq", default="00")
    ap.add_argument("--write-bag")
    ap.add_argument("--reference-order", action="store_true", help="sum voxel members in pcl::VoxelGrid's own order (the reference's bits; ~4x slower for one sensor): for runs that are compared pose by pose with A-LOAM's")
    ap.add_argument("--selftest", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if args.from_kitti:
        spec = importlib.util.spec_from_file_location("run_kitti", os.path.join(ROOT, "tools", "run_kitti.py"))
        kitti = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(kitti)
        times = kitti.read_times(os.path.join(args.from_kitti, "sequences", args.seq, "times.txt"))
        if args.max_frames:
            times = times[: args.max_frames]
        clouds = (kitti.read_lidar(os.path.join(args.from_kitti, "velodyne", "sequences", args.seq, "velodyne", f"{k:06d}.bin")) for k in range(len(times)))
        write_bag(args.write_bag or os.path.join(args.out, f"kitti_{args.seq}.bag"), clouds, times)
        return
    if args.selftest:
        syn = importlib.import_module("a-loam_amd.synthetic")
        scans, _, _, model = syn.make_sequence("VLP-16", 5, seed=5)
        args.bag = os.path.join(args.out, "selftest.bag")
        args.scan_line, args.minimum_range = model.n_scans, model.min_range
        write_bag(args.bag, [s.numpy() for s in scans], [0.1 * k for k in range(len(scans))], compression="bz2")
    binding = importlib.import_module("a-loam_amd.binding")
    gpu = binding.Aloam(n_scans=args.scan_line, min_range=args.minimum_range, max_points=400000)
    if args.reference_order:
        gpu.set_voxel_sum_order(True)
    if args.mapping:
        gpu.mapping_enable(args.line_res, args.plane_res, pool_points=1 << 17)   # grows with the map
    odo, mapped = [], []
    for k, (stamp, xyz) in enumerate(sweeps(args.bag, args.topic)):
        if args.max_frames and k >= args.max_frames:
            break
        gpu.scan_register(xyz)                                   # 12-byte records
        gpu.odometry_step()
        p = gpu.pose()
        odo.append([stamp, *p["t_w"], *p["q_w"]])
        if args.mapping:
            gpu.mapping_step()
            gpu.synchronize()
            m = gpu.map_pose()
            mapped.append([stamp, *m["t_w"], *m["q_w"]])
    if not odo:
        raise SystemExit(f"no {rosbag1.POINTCLOUD2_TYPE} messages on {args.topic} in {args.bag}")
    np.savetxt(os.path.join(args.out, "odometry.txt"), np.array(odo), fmt="%.9e")
    if mapped:
        np.savetxt(os.path.join(args.out, "mapped.txt"), np.array(mapped), fmt="%.9e")
    print(f"{len(odo)} sweeps from {args.bag} -> {args.out}/odometry.txt" + (" + mapped.txt" if mapped else ""))
    gpu.close()


if __name__ == "__main__":
    main()
