import sys, importlib, os, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+"/oracle")
binding = importlib.import_module("a-loam_amd.binding")
syn = importlib.import_module("a-loam_amd.synthetic")
scans, R, t, model = syn.make_sequence("VLP-16", 3, seed=3)
gpu = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, batch=1, max_points=40000, device=0)
gpu.mapping_enable(0.2, 0.4, pool_points=65536)
for s in scans:
    gpu.scan_register(s.numpy()); gpu.odometry_step(); gpu.mapping_step(); gpu.synchronize()
    print(gpu.map_info(), flush=True)
