"""oracle/ref_py.py — runs the oracle/_ref executables (the reference's own translation units compiled against the
stand-in third-party headers of oracle/ref_shim/) on numpy inputs.  TEST INFRASTRUCTURE ONLY: used by tests/ to check the
oracle's restatement against the reference's code, and by tools/make_golden.py to generate tests/golden/ref_*.npz.

The executables can only be (re)built where /root/reference exists (`make -C oracle _ref`); prebuilt ones travel with
the repo snapshot to the GPU box (oracle/_ref/ is git-ignored but not gpurun-ignored)."""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
REFERENCE_ROOT = os.environ.get("ALOAM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, n)) for n in ("ref_scan_registration", "ref_laser_odometry", "ref_laser_mapping"))


def build() -> bool:
    """Build oracle/_ref if the reference sources are present; returns whether the executables exist afterwards."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        r = subprocess.run(["make", "-C", _HERE, "_ref", f"REF={REFERENCE_ROOT}"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building oracle/_ref failed:\n" + r.stdout + r.stderr)
    return available()


def _write_cloud(f, a):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
    f.write(struct.pack("<i", a.shape[0]))
    f.write(a.tobytes())


class _Reader:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        self.o = 0

    def i32(self):
        v = struct.unpack_from("<i", self.b, self.o)[0]
        self.o += 4
        return v

    def f64(self, n):
        v = np.frombuffer(self.b, dtype="<f8", count=n, offset=self.o).copy()
        self.o += 8 * n
        return v

    def cloud(self):
        n = self.i32()
        v = np.frombuffer(self.b, dtype="<f4", count=4 * n, offset=self.o).reshape(n, 4).copy()
        self.o += 16 * n
        return v

    def done(self):
        return self.o == len(self.b)


def _timing(stderr):
    """`REF_TIMING <stage> frames <n> seconds <s>` lines of a driver run with REF_TIMING=1 -> seconds inside the reference's own code."""
    for line in stderr.splitlines():
        if line.startswith("REF_TIMING"):
            return float(line.split()[-1])
    return None


LAST_TIMING = {}     # stage -> seconds the reference's own code spent in the last timed run (timing=True)


def scan_registration(scans, n_scans, min_range, exe=None, timing=False):
    """scans: list of (n, 4) float32 arrays -> list of dicts with the five published clouds + curvature / label / picked."""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", len(scans)))
            for s in scans:
                _write_cloud(f, s)
        r = subprocess.run([exe or os.path.join(REF_DIR, "ref_scan_registration"), str(int(n_scans)), repr(float(min_range)), fin, fout],
                           capture_output=True, text=True, env=dict(os.environ, REF_TIMING="1") if timing else None)
        if timing:
            LAST_TIMING["scan_registration"] = _timing(r.stderr)
        if r.returncode != 0:
            raise RuntimeError(f"ref_scan_registration failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
        rd = _Reader(fout)
        out = []
        for _ in scans:
            fr = {k: rd.cloud() for k in ("cloud", "sharp", "less_sharp", "flat", "less_flat")}
            n = rd.i32()
            rec = np.frombuffer(rd.b, dtype=np.dtype([("c", "<f4"), ("l", "<i4"), ("p", "<i4")]), count=n, offset=rd.o)
            rd.o += 12 * n
            fr["curvature"], fr["label"], fr["picked"] = rec["c"].copy(), rec["l"].copy(), rec["p"].copy()
            out.append(fr)
        assert rd.done()
        return out


def laser_odometry(frames, exe=None, exe_args=(), timing=False):
    """frames: list of dicts with sharp / less_sharp / flat / less_flat / cloud -> list of dicts (q_w, t_w, q_lc, t_lc,
    corner_corr, plane_corr, corner_last, surf_last), one per frame, from ONE run of the node (state carries over)."""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", len(frames)))
            for fr in frames:
                for k in ("sharp", "less_sharp", "flat", "less_flat", "cloud"):
                    _write_cloud(f, fr[k])
        r = subprocess.run([exe or os.path.join(REF_DIR, "ref_laser_odometry"), *[str(a) for a in exe_args], fin, fout], capture_output=True, text=True,
                           env=dict(os.environ, REF_TIMING="1") if timing else None)
        if timing:
            LAST_TIMING["laser_odometry"] = _timing(r.stderr)
        if r.returncode != 0:
            raise RuntimeError(f"ref_laser_odometry failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
        rd = _Reader(fout)
        out = []
        for _ in frames:
            v = rd.f64(14)
            cc, pc = rd.i32(), rd.i32()
            fr = {"q_w": v[0:4], "t_w": v[4:7], "q_lc": v[7:11], "t_lc": v[11:14], "corner_corr": cc, "plane_corr": pc,
                  "corner_last": rd.cloud(), "surf_last": rd.cloud()}
            ne, npl = rd.i32(), rd.i32()                        # correspondences of the frame's last ceres::Solve (constructor arguments of the factors)
            fr["edges"] = rd.f64(9 * ne).reshape(ne, 9)
            fr["planes"] = rd.f64(12 * npl).reshape(npl, 12)
            out.append(fr)
        assert rd.done()
        return out


def laser_mapping(frames, line_res, plane_res, dump_map=True, exe=None, exe_args=()):
    """frames: list of dicts with q_w / t_w (odometry pose), corner_last, surf_last, cloud (full resolution) -> list of dicts
    (q_w, t_w = refined pose, q_wmap_wodom, t_wmap_wodom, registered, cen, corner_map {cube: pts}, surf_map {cube: pts})."""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", len(frames)))
            for fr in frames:
                f.write(np.concatenate([np.asarray(fr["q_w"], np.float64), np.asarray(fr["t_w"], np.float64)]).tobytes())
                for k in ("corner_last", "surf_last", "cloud"):
                    _write_cloud(f, fr[k])
        r = subprocess.run([exe or os.path.join(REF_DIR, "ref_laser_mapping"), *[str(a) for a in exe_args], repr(float(line_res)), repr(float(plane_res)), fin, fout],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"ref_laser_mapping failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
        rd = _Reader(fout)
        out = []
        for _ in frames:
            v = rd.f64(14)
            fr = {"q_w": v[0:4], "t_w": v[4:7], "q_wmap_wodom": v[7:11], "t_wmap_wodom": v[11:14], "registered": rd.cloud(),
                  "cen": (rd.i32(), rd.i32(), rd.i32())}
            for name in ("corner_map", "surf_map"):
                m = {}
                for _c in range(rd.i32()):
                    cube = rd.i32()
                    m[cube] = rd.cloud()
                fr[name] = m if dump_map else {c: len(p) for c, p in m.items()}
            out.append(fr)
        assert rd.done()
        return out


def lidar_factors(records, exe=None):
    """records: (n, 21) float64 rows [kind, s, q(xyzw), t, 12 constants] -> residual (n,3), d r/d q (n,3,4), d r/d t (n,3,3)
    from the reference's LidarEdgeFactor / LidarPlaneFactor through ceres::AutoDiffCostFunction (stand-in Ceres)."""
    rec = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 21)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", rec.shape[0]))
            f.write(rec.tobytes())
        subprocess.run([exe or os.path.join(REF_DIR, "ref_lidar_factor"), fin, fout], check=True)
        out = np.fromfile(fout, dtype="<f8").reshape(rec.shape[0], 24)
    return out[:, :3].copy(), out[:, 3:15].reshape(-1, 3, 4).copy(), out[:, 15:].reshape(-1, 3, 3).copy()
