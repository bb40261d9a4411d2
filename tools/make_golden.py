"""Regenerates tests/golden/*.npz: oracle outputs on small seeded synthetic sweeps.

These goldens are SELF-GENERATED (the reference ships no fixtures and cannot be built here): they pin the oracle
against regressions and give the GPU tests fixed vectors, they do not pin the oracle to the reference.
    python tools/make_golden.py
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
syn = importlib.import_module("a-loam_amd.synthetic")

CASES = [("vlp16_c600_seed5", "VLP-16", 3, 5, {"columns": 600}), ("hdl64_c256_seed6", "HDL-64", 3, 6, {"columns": 256})]


def main():
    for tag, name, frames, seed, kw in CASES:
        scans, R, t, model = syn.make_sequence(name, frames, seed=seed, **kw)
        orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
        out = {"R": R.numpy(), "t": t.numpy(), "n_scans": model.n_scans, "min_range": model.min_range}
        for k, s in enumerate(scans):
            x = s.numpy()
            f = orc.scan_register(x)
            p = orc.odometry_step()
            out[f"scan{k}"] = x
            for key in ("sharp", "less_sharp", "flat", "less_flat"):
                out[f"{key}{k}"] = f[key]
            out[f"cloud_intensity{k}"] = f["cloud"][:, 3].copy()
            for key in ("q_lc", "t_lc", "q_w", "t_w"):
                out[f"{key}{k}"] = p[key]
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
