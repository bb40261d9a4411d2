"""Which sequences of bench.py's travelling workload leave the pair kernel (flags of k_build_grids_fused)?  GPU box only."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
binding = importlib.import_module("a-loam_amd.binding"); syn = importlib.import_module("a-loam_amd.synthetic")
dev = torch.device("cuda", 0)
wl = bench.TravelWorkload(syn, torch, 32, 12, 0, dev)
cx = wl.ctx(binding, 32, 0)
base = wl.data.data_ptr()
for k in range(12):
    cx.process_device(base + k * wl.NP * 16, wl.seq_stride, wl.nin(k))
    cx.synchronize()
    bad = []
    for b in range(32):
        for which, nm in ((binding.CLOUD_CORNER_LAST, "corner"), (binding.CLOUD_SURF_LAST, "surf")):
            c = cx.cloud(which, b)
            key = c[:, 3].astype(np.int32)
            d = int((np.diff(key) < 0).sum())
            if d or key.min() < 0 or np.abs(c[:, :3]).max() >= 4096:
                i = int(np.argmax(np.diff(key) < 0))
                bad.append((b, nm, d, int((np.maximum.accumulate(key) - key).max()), c[max(0, i - 1):i + 3, 3].tolist()))
    print("frame", k, "flagged:", bad[:6], "..." if len(bad) > 6 else "", len(bad))
