#!/usr/bin/env python
"""What if the third-party code the reference links (absent here: PCL 1.8, Eigen 3, FLANN) behaves differently from this repo's stand-ins in the two
places nobody could settle from the sources at hand?

  recip : pcl::VoxelGrid's centroid scales the sums by 1 / n (one rounding more) instead of dividing them - Eigen's vector / scalar has been either,
          depending on the version;
  ties  : nearestKSearch resolves exactly equal distances towards the HIGHER index (FLANN returns them in traversal order; the stand-in, the oracle and
          the device take the lower index).

Each variant is the reference's OWN three translation units (oracle/_ref recipe, one -D on the stand-in header) run end to end over the 300-frame
drive of tests/golden/reflong_*.npz and compared with the fixture (the same units, default stand-ins).  Needs /root/reference; writes
profiles/r06_third_party_sensitivity.json.

    python tools/third_party_sensitivity.py
"""
import importlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_py  # noqa: E402
from conftest import quat_angle  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle")
REF = ref_py.REFERENCE_ROOT
VARIANTS = {"recip": "-DSHIM_CENTROID_BY_RECIPROCAL", "ties": "-DSHIM_KNN_TIES_HIGHEST"}


def build(name, define):
    out = os.path.join(ORACLE, "_ref", "var_" + name)
    os.makedirs(out, exist_ok=True)
    flags = f"-O3 -std=c++14 -ffp-contract=off -w -Iref_shim/include -Iref_shim -I{REF}/include -I{REF}/src -Dmain=ref_node_main {define}".split()
    for tu, drv, extra in (("scanRegistration", "driver_scan_registration", []), ("laserOdometry", "driver_laser_odometry", []),
                           ("laserMapping", "driver_laser_mapping", ["-include", "ref_shim/mapping_prefix.hpp"])):
        obj = os.path.join(out, tu + ".o")
        subprocess.run(["g++", *flags, *extra, "-c", f"{REF}/src/{tu}.cpp", "-o", obj], cwd=ORACLE, check=True)
        subprocess.run(["g++", *flags, "-Umain", f"ref_shim/{drv}.cpp", obj, "-o", os.path.join(out, "ref_" + drv[len("driver_"):]), "-lpthread"], cwd=ORACLE, check=True)
    return out


def main():
    assert os.path.isdir(REF), "the reference sources are needed"
    syn = importlib.import_module("a-loam_amd.synthetic")
    g = np.load(os.path.join(ROOT, "tests", "golden", "reflong_hdl64_c512_seed51.npz"))
    frames = int(g["frames"])
    scans, _, _, model = syn.make_sequence(str(g["sensor"]), frames, seed=int(g["seed"]), **json.loads(str(g["kwargs"])))
    xs = [s.numpy() for s in scans]
    env = np.maximum.accumulate(np.linalg.norm(g["ulp_t_w"] - g["t_w"], axis=1))
    report = {"fixture": "tests/golden/reflong_hdl64_c512_seed51.npz", "frames": frames,
              "one_ulp_envelope_m": {"frame_25": float(env[25]), "frame_100": float(env[100]), "max": float(env[-1])}, "variants": {}}
    for name, define in VARIANTS.items():
        d = build(name, define)
        reg = ref_py.scan_registration(xs, model.n_scans, model.min_range, exe=os.path.join(d, "ref_scan_registration"))
        odo = ref_py.laser_odometry(reg, exe=os.path.join(d, "ref_laser_odometry"))
        fr = [dict(q_w=o["q_w"], t_w=o["t_w"], corner_last=o["corner_last"], surf_last=o["surf_last"], cloud=r["cloud"]) for o, r in zip(odo, reg)]
        mp = ref_py.laser_mapping(fr, float(g["line_res"]), float(g["plane_res"]), dump_map=False, exe=os.path.join(d, "ref_laser_mapping"))
        dt = np.array([np.linalg.norm(m["t_w"] - g["t_w"][k]) for k, m in enumerate(mp)])
        dr = np.array([quat_angle(m["q_w"], g["q_w"][k]) for k, m in enumerate(mp)])
        do = np.array([np.linalg.norm(o["t_w"] - g["odom_t"][k]) for k, o in enumerate(odo)])
        first = int(np.argmax(dt > 1e-4)) if (dt > 1e-4).any() else -1
        report["variants"][name] = {"define": define, "refined_pose_diff_m": {"frame_25": float(dt[25]), "frame_100": float(dt[100]), "max": float(dt.max())},
                                    "refined_rotation_diff_rad_max": float(dr.max()), "odometry_chain_diff_m_max": float(do.max()),
                                    "first_frame_beyond_1e-4_m": first, "identical_to_fixture": bool(dt.max() == 0.0 and do.max() == 0.0)}
        print(name, json.dumps(report["variants"][name]))
    out = os.path.join(ROOT, "profiles", "r06_third_party_sensitivity.json")
    json.dump(report, open(out, "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
