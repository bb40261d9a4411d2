import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
binding = importlib.import_module("a-loam_amd.binding")
syn = importlib.import_module("a-loam_amd.synthetic")
scans, R, t, model = syn.make_sequence("ROWS128", 1, seed=1)
x = scans[0].numpy()
orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
gpu = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field, max_points=len(x) + 64)
fo = orc.scan_register(x); gpu.scan_register(x)
co, lo, pk = orc.per_point(); cg, lg = gpu.per_point()
so, cnt = orc.ring_ranges()
bad = np.nonzero(lo != lg)[0]
print("label mismatches", len(bad), bad[:10])
for i in bad[:3]:
    r = np.searchsorted(so, i, side="right") - 1
    n = cnt[r]; L = n - 11; e = i - so[r] - 5
    sec = max(j for j in range(6) if (L * j) // 6 <= e)
    sp = (L * sec) // 6; ep = (L * (sec + 1)) // 6 - 1
    print("idx", i, "ring", r, "n", n, "local", i - so[r], "e", e, "sector", sec, "sp", sp, "ep", ep, "pos in sector", e - sp, "label oracle/gpu", lo[i], lg[i], "curv", co[i])
    lo_s = lo[so[r] + 5 + sp: so[r] + 5 + ep + 1]; lg_s = lg[so[r] + 5 + sp: so[r] + 5 + ep + 1]
    print("  sector labels oracle nonzero:", [(int(k), int(v)) for k, v in enumerate(lo_s) if v != 0][:40])
    print("  sector labels gpu    nonzero:", [(int(k), int(v)) for k, v in enumerate(lg_s) if v != 0][:40])
    prev = lo[so[r] + 5 + max(0, sp - 8): so[r] + 5 + sp]
    print("  prev sector tail labels (oracle):", prev.tolist(), "picked oracle at head:", pk[so[r] + 5 + sp: so[r] + 5 + sp + 8].tolist())
    cs = co[so[r] + 5 + sp: so[r] + 5 + ep + 1]
    print("  n with c>0.1:", int((cs > 0.1).sum()), "top curv idx:", np.argsort(-cs)[:6].tolist(), cs[np.argsort(-cs)[:6]].tolist())
