"""KITTI-odometry input formats and ground-truth frame convention (reference src/kittiHelper.cpp:25-35,68-134),
as read by tools/run_kitti.py.  KITTI itself is not in the image: the layout is written synthetically and read back."""
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("run_kitti", os.path.join(ROOT, "tools", "run_kitti.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_ground_truth_frame_rotation_hand_computed(tmp_path):
    """R_transform = [0 0 1; -1 0 0; 0 -1 0] (kittiHelper.cpp:78-80) applied as q_transform * q_w_i and q_transform * t (:104-107)."""
    k = _tool()
    c, s = np.cos(0.3), np.sin(0.3)
    Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])                      # camera frame: yaw about the camera's y axis
    poses = [np.hstack([np.eye(3), [[1.0], [2.0], [3.0]]]), np.hstack([Ry, [[0.5], [-0.25], [10.0]]])]
    path = tmp_path / "00.txt"
    path.write_text("\n".join(" ".join(f"{v:.9e}" for v in P.reshape(-1)) for P in poses) + "\n")
    R, t = k.read_gt(str(path))
    assert R.shape == (2, 3, 3) and t.shape == (2, 3)
    assert np.allclose(t[0], [3.0, -1.0, -2.0], atol=1e-6)                 # (x, y, z)_cam -> (z, -x, -y)
    assert np.allclose(R[0], [[0, 0, 1], [-1, 0, 0], [0, -1, 0]], atol=1e-7)
    assert np.allclose(t[1], [10.0, -0.5, 0.25], atol=1e-6)
    assert np.allclose(R[1], k.R_TRANSFORM @ Ry, atol=1e-6)
    # entries go through stof (single precision), like the reference
    path.write_text("1 0 0 0.123456789012 0 1 0 0 0 0 1 0\n")
    _, t1 = k.read_gt(str(path))
    assert t1[0, 2] == -0.0 or t1[0, 2] == 0.0
    assert t1[0, 1] == -float(np.float32(0.123456789012))


def test_layout_round_trip(tmp_path):
    """sequences/<seq>/times.txt, velodyne/sequences/<seq>/velodyne/%06d.bin (float32 x 4), results/<seq>.txt (kittiHelper.cpp:68-72,
    127-134): what write_selftest lays down is what the readers return, and the ground truth is the synthetic trajectory."""
    k = _tool()
    folder = str(tmp_path / "ds")
    k.write_selftest(folder, "07", frames=3)
    times = k.read_times(os.path.join(folder, "sequences", "07", "times.txt"))
    assert np.allclose(times, [0.0, 0.1, 0.2], atol=1e-7)
    syn = importlib.import_module("a-loam_amd.synthetic")
    scans, R, t, model = syn.make_sequence("HDL-64", 3, seed=77, columns=1024)
    for i, s in enumerate(scans):
        pts = k.read_lidar(os.path.join(folder, "velodyne", "sequences", "07", "velodyne", f"{i:06d}.bin"))
        assert pts.dtype == np.float32 and pts.shape == tuple(s.shape) and np.array_equal(pts.view(np.uint32), s.numpy().view(np.uint32))
    Rg, tg = k.read_gt(os.path.join(folder, "results", "07.txt"))
    Rn, tn = R.numpy(), t.numpy()
    for i in range(3):                                                      # lidar pose relative to the first sweep
        assert np.allclose(Rg[i], Rn[0].T @ Rn[i], atol=1e-5) and np.allclose(tg[i], Rn[0].T @ (tn[i] - tn[0]), atol=1e-4)


@pytest.mark.gpu
def test_selftest_run_with_mapping(tmp_path):
    """tools/run_kitti.py --selftest --mapping on the GPU box: the whole read -> register -> odometry -> mapping -> ATE path."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_kitti.py"), "--selftest", "--mapping", "--out", str(tmp_path / "out")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ate = {m.group(1): float(m.group(2)) for m in re.finditer(r"(odometry|mapped): \d+ sweeps, ATE \(RMSE, no alignment\) = ([0-9.]+) m", r.stdout)}
    assert set(ate) == {"odometry", "mapped"}, r.stdout
    assert ate["odometry"] < 0.5 and ate["mapped"] < 0.05 and ate["mapped"] < ate["odometry"], ate   # scan-to-scan drifts ~5 cm per sweep on this world (the oracle does too); the map refinement removes it
    for name in ("00_odometry.txt", "00_mapped.txt"):
        tr = np.loadtxt(tmp_path / "out" / name)
        assert tr.shape == (6, 8) and np.allclose(np.linalg.norm(tr[:, 4:], axis=1), 1.0, atol=1e-9)
