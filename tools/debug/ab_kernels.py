import numpy as np, json, sys
p=json.loads(str(np.load("gpurun_out/x.npz")["profile"]))["ms_per_step"]
keys = sys.argv[1:] or ["k_find_ends","k_front","k_ring_starts","k_ring_features","k_build_grids"]
print("   ", {k: round(v,3) for k,v in p.items() if any(k.startswith(q) for q in keys)}, "sum", round(sum(p.values()),3))
