"""Ad-hoc GPU-vs-oracle comparison with verbose output (development aid; the real checks live in tests/)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
syn = importlib.import_module("a-loam_amd.synthetic")
bind = importlib.import_module("a-loam_amd.binding")
import oracle_py as O


def cmp(name, a, b):
    if a.shape != b.shape:
        print(f"  {name}: SHAPE {a.shape} vs {b.shape}")
        n = min(len(a), len(b))
        if n:
            bad = np.nonzero((a[:n] != b[:n]).reshape(n, -1).any(axis=1))[0]
            print(f"     first mismatch row {bad[:5]} of common {n}")
        return False
    eq = np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    if eq:
        print(f"  {name}: bit-exact ({a.shape})")
    else:
        bad = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0]
        print(f"  {name}: {len(bad)} rows differ of {len(a)}; first {bad[:5]}; maxabs {np.nanmax(np.abs(a.astype(np.float64)-b.astype(np.float64)))}")
        for r in bad[:3]:
            print("     ", r, a[r], b[r])
    return eq


def run(name, frames=4, seed=1):
    scans, R, t, model = syn.make_sequence(name, frames, seed=seed)
    o = O.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
    g = bind.Aloam(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field, batch=1, max_points=140000 if name != "ROWS128" else 270000)
    for k, s in enumerate(scans):
        x = s.numpy()
        t0 = time.time(); fo = o.scan_register(x); t1 = time.time()
        g.scan_register(x); fg = g.features(); t2 = time.time()
        print(f"[{name} frame {k}] oracle reg {1e3*(t1-t0):.1f} ms; gpu reg+readback {1e3*(t2-t1):.1f} ms")
        so, co = o.ring_ranges(); sg, cg = g.ring_ranges()
        cmp("ring_start", so, sg); cmp("ring_count", co, cg)
        for key in ("cloud", "sharp", "less_sharp", "flat", "less_flat"):
            cmp(key, fo[key], fg[key])
        curv_o, lab_o, _ = o.per_point(); curv_g, lab_g = g.per_point()
        # compare only where the reference consumes them: ring interior
        mask = np.zeros(len(curv_o), bool)
        for s0, c0 in zip(so, co):
            if c0 >= 17: mask[s0 + 5:s0 + c0 - 6] = True
        cmp("curvature(sel)", curv_o[mask], curv_g[mask]); cmp("label(sel)", lab_o[mask], lab_g[mask])
        po = o.odometry_step(); g.odometry_step(); pg = g.pose()
        for key in ("q_lc", "t_lc", "q_w", "t_w"):
            print(f"  pose {key}: oracle {po[key]} gpu {pg[key]} |d|={np.abs(po[key]-pg[key]).max():.3e}")
        print("  stats oracle", o.odom_stats()); print("  stats gpu   ", g.odom_stats())
    g.close()


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["VLP-16", "HDL-64"]):
        run(nm)
