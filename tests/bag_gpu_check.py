"""Helper of tests/test_bag_io.py (GPU leg) and a stand-alone check: every /velodyne_points sweep of a ROS 1 bag goes through
tools/run_bag.py's reader into the C ABI as 12-byte x y z records and through the CPU oracle as the same points; features must be
bit-identical and poses within 1e-4 (test infrastructure: this file may use oracle/).
    python tests/bag_gpu_check.py <bag> <scan_line> <minimum_range>"""
import importlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def check(bag, scan_line, min_range, topic="/velodyne_points"):
    spec = importlib.util.spec_from_file_location("run_bag", os.path.join(ROOT, "tools", "run_bag.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    binding = importlib.import_module("a-loam_amd.binding")
    import oracle_py
    gpu = binding.Aloam(n_scans=scan_line, min_range=min_range, max_points=200000)
    orc = oracle_py.Oracle(n_scans=scan_line, min_range=min_range)
    n = 0
    for stamp, xyz in tool.sweeps(bag, topic):
        fo = orc.scan_register(np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1))   # the 4th input float is never read (scanRegistration.cpp:132-133)
        gpu.scan_register(xyz)
        fg = gpu.features()
        for k in ("cloud", "sharp", "less_sharp", "flat", "less_flat"):
            assert fo[k].shape == fg[k].shape and np.array_equal(fo[k].view(np.uint32), fg[k].view(np.uint32)), (n, k)
        po = orc.odometry_step()
        gpu.odometry_step()
        pg = gpu.pose()
        assert np.abs(po["t_w"] - pg["t_w"]).max() < 1e-4 and np.abs(po["q_w"] - pg["q_w"]).max() < 1e-4, (n, po, pg)
        n += 1
    gpu.close()
    return n


if __name__ == "__main__":
    print("BAG CHECK OK:", check(sys.argv[1], int(sys.argv[2]), float(sys.argv[3])), "sweeps")
