"""Correspondences as INDICES: the reference's factors, the oracle's and the HIP path's records hold the coordinates of the points a
feature was associated with (laserOdometry.cpp:365-381, 460-479); this turns them into (query index, closestPointInd, minPointInd2[,
minPointInd3]) by exact coordinate look-up in the clouds they were taken from, so that two runs can be compared index by index even when
their less-flat centroids differ in the last bits (each run is looked up in its OWN clouds; the clouds have the same order and size)."""
import numpy as np


def _lookup(cloud):
    xyz = np.ascontiguousarray(np.asarray(cloud)[:, :3], np.float32)
    keys = xyz.view([("k", "V12")]).ravel()
    first = {}
    for i, k in enumerate(keys):
        first.setdefault(k["k"].tobytes(), i)
    return first


def indices(records, query_cloud, target_cloud):
    """records [n, 3 + 3 m] (current point, then m associated points) -> int array [n, 1 + m]; -1 where a point is not in its cloud."""
    rec = np.ascontiguousarray(np.asarray(records, np.float64).astype(np.float32))
    n, m = rec.shape[0], rec.shape[1] // 3 - 1
    out = np.full((n, 1 + m), -1, np.int64)
    if n == 0:
        return out
    q, t = _lookup(query_cloud), _lookup(target_cloud)
    for i in range(n):
        out[i, 0] = q.get(rec[i, 0:3].tobytes(), -1)
        for j in range(m):
            out[i, 1 + j] = t.get(rec[i, 3 + 3 * j:6 + 3 * j].tobytes(), -1)
    return out


def compare(a, b):
    """Two index tables (rows keyed by their query index) -> dict: queries in both, rows that differ, queries only in one."""
    da, db = {int(r[0]): tuple(int(v) for v in r[1:]) for r in a}, {int(r[0]): tuple(int(v) for v in r[1:]) for r in b}
    both = sorted(set(da) & set(db))
    diff = [q for q in both if da[q] != db[q]]
    return {"both": len(both), "differ": len(diff), "only_a": len(set(da) - set(db)), "only_b": len(set(db) - set(da)), "which": diff}
