"""One CPU-baseline worker: runs the oracle (oracle/, the CPU restatement of the reference — test infrastructure, here only as
bench.py's `cpu_baseline` leg) over the sweeps of one synthetic sequence for a bounded time and prints one JSON line.

    python tools/cpu_baseline_worker.py <sweeps.npz> <seconds> [--mapping]

bench.py starts one of these per host core for the `nproc`-way sequence-parallel figure (each reference node is
single-threaded, so the fair whole-host number is one independent sequence per core).  Imports numpy + ctypes only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np


def main():
    path, seconds = sys.argv[1], float(sys.argv[2])
    mapping = "--mapping" in sys.argv[3:]
    import oracle_py
    g = np.load(path)
    sweeps = [g[f"s{k}"] for k in range(int(g["T"]))]
    order = [int(v) for v in g["order"]]
    orc = oracle_py.Oracle(n_scans=int(g["n_scans"]), min_range=float(g["min_range"]), ring_from_field=bool(g["ring_from_field"]))
    if mapping:
        orc.map_config(float(g["line_res"]), float(g["plane_res"]))
    per = []
    t_end = time.perf_counter() + seconds
    i = 0
    while time.perf_counter() < t_end or len(per) < 3:
        x = sweeps[order[i % len(order)]]
        i += 1
        t0 = time.perf_counter()
        orc.scan_register(x)
        po = orc.odometry_step()
        if mapping:
            orc.mapping_step(po["q_w"], po["t_w"], orc.cloud(oracle_py.CLOUD_CORNER_LAST), orc.cloud(oracle_py.CLOUD_SURF_LAST), orc.cloud(oracle_py.CLOUD_FULL))
        per.append(time.perf_counter() - t0)
    print(json.dumps({"scans": len(per), "seconds": float(sum(per)), "per_scan_ms": [round(1e3 * v, 3) for v in per]}))


if __name__ == "__main__":
    main()
