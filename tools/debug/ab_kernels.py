import numpy as np, json, sys
p=json.loads(str(np.load("gpurun_out/x.npz")["profile"]))["ms_per_step"]
print("   ", {k: round(v,3) for k,v in p.items() if k in ("k_find_ends","k_front","k_ring_starts","k_classify","k_scatter","k_ring_offsets","k_ring_features","k_build_grids")}, "sum", round(sum(p.values()),3))
