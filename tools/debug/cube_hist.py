"""Sizes of the map cubes the per-cube re-filter works on at steady-state map depth (bench.py's travelling workload).  GPU box only."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
binding = importlib.import_module("a-loam_amd.binding"); syn = importlib.import_module("a-loam_amd.synthetic")
dev = torch.device("cuda", 0)
B, T = 16, 100
wl = bench.TravelWorkload(syn, torch, B, T, 0, dev, distinct=16)
cx = wl.ctx(binding, B, 0)
cx.mapping_enable(0.4, 0.8, 262144)
base = wl.data.data_ptr()
order = bench.frame_order(T, 140)
for i, k in enumerate(order):
    cx.process_device(base + k * wl.NP * 16, wl.seq_stride, wl.nin(k))
    cx.mapping_step()
    if i in (79, 139):
        cx.synchronize()
        for b in (0, 5):
            for cls in (0, 1):
                cnt = np.zeros(21 * 21 * 11, np.int32)
                binding.lib().aloam_map_cube_counts(cx.h, b, cls, binding._p(cnt))
                nz = np.sort(cnt[cnt > 0])[::-1]
                print(f"step {i} seq {b} {'corner' if cls == 0 else 'surf'}: {len(nz)} cubes, {int(nz.sum())} points; > 8192: {int((nz > 8192).sum())}, 2049..8192: {int(((nz > 2048) & (nz <= 8192)).sum())}, <= 2048: {int((nz <= 2048).sum())}; sizes {nz.tolist()}")
        print(cx.map_info(0))
