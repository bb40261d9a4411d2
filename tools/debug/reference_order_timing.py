"""Milliseconds per sweep of the reference-order validation mode against the default order (batch 1, 64 x 2048 and 64 x 512).  GPU box only."""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
binding = importlib.import_module("a-loam_amd.binding"); syn = importlib.import_module("a-loam_amd.synthetic")
for cols in (512, 2048):
    scans, R, t, model = syn.make_sequence("HDL-64", 30, seed=5, columns=cols, travel=True, step=1.6)
    xs = [s.numpy() for s in scans]
    for ref in (False, True):
        gpu = binding.Aloam(n_scans=64, min_range=5.0, max_points=max(len(x) for x in xs) + 256)
        gpu.set_voxel_sum_order(ref)
        gpu.mapping_enable(0.4, 0.8, pool_points=131072)
        tr = to = tm = 0.0
        for k, x in enumerate(xs):
            t0 = time.perf_counter(); gpu.scan_register(x); t1 = time.perf_counter(); gpu.odometry_step(); gpu.synchronize(); t2 = time.perf_counter(); gpu.mapping_step(); gpu.synchronize(); t3 = time.perf_counter()
            if k >= 10: tr += t1 - t0; to += t2 - t1; tm += t3 - t2
        n = len(xs) - 10
        print(f"64 x {cols} {'reference order' if ref else 'input order    '}: registration {1e3*tr/n:.2f} ms, odometry {1e3*to/n:.2f} ms, mapping {1e3*tm/n:.2f} ms per sweep (batch 1)")
        gpu.close()
