"""Model check of the association search the HIP kernel performs (a-loam_amd/csrc/odometry_kernels.hip, k_associate).

Not the kernel itself — a line-by-line Python model of its search STRUCTURE (hash buckets with collisions, f32 cell
binning, the fine 3x3x3 block, coarse shells pruned by the best distance, the 1-NN block's candidates re-used for the
ring-neighbour classes, the single-level ring grid with its 3x3 and 5x5 blocks, the early-out bounds and the 1 mm / 0.1 %
pruning margins) run against the plain definition of what the reference computes (src/laserOdometry.cpp:299-483: exact
nearest neighbour, then the walk-until-break loops).  The GPU tests compare the real kernel with the oracle on a handful
of sweeps; this runs the same decision logic over many random clouds on the CPU, so an unsound bound or margin shows up
without a GPU."""
import numpy as np
import pytest

F = np.float32
IDX_MASK = (1 << 20) - 1


def _hash3(a, b, c, H):
    m = 0xFFFFFFFF
    return (((a & m) * 73856093 & m) ^ ((b & m) * 19349663 & m) ^ ((c & m) * 83492791 & m)) & (H - 1)


def _cell(v, cell):
    return int(np.floor(F(v) * (F(1.0) / F(cell))))


class Grid:
    """Counting-sort buckets like k_build_grids: an entry is (x, y, z, idx, key)."""

    def __init__(self, pts, keys, cell, H, third_is_key):
        self.cell, self.H = cell, H
        self.b = [[] for _ in range(H)]
        for i, (p, k) in enumerate(zip(pts, keys)):
            c = (_cell(p[0], cell), _cell(p[1], cell), int(k) if third_is_key else _cell(p[2], cell))
            self.b[_hash3(c[0], c[1], c[2], H)].append((p, i, int(k)))

    def bucket(self, a, b, c):
        return self.b[_hash3(a, b, c, self.H)]


def _d2(p, s):
    dx, dy, dz = F(p[0]) - F(s[0]), F(p[1]) - F(s[1]), F(p[2]) - F(s[2])
    return F(F(dx * dx + dy * dy) + dz * dz)


def _gap(s, c, cell):
    lo = F(c) * F(cell); hi = F(lo + F(cell)); s = F(s)
    g = F(lo - s) if s < lo else (F(s - hi) if s > hi else F(0))
    return F(g - F(1e-3)) if g > F(1e-3) else F(0)


def model_query(sel, g3, g3c, g2, plane, sweep_rows):
    """-> (closest, min2, min3) as the kernel would pick them, or None."""
    cell3, cellc, cell2 = g3.cell, g3c.cell, g2.cell
    best = None                                                              # (d, idx, key)
    def take(cur, d, idx, key):
        return (d, idx, key) if cur is None or (d, idx) < (cur[0], cur[1]) else cur
    cx, cy, cz = _cell(sel[0], cell3), _cell(sel[1], cell3), _cell(sel[2], cell3)
    fine = []
    for l in range(27):
        fine += g3.bucket(cx + l % 3 - 1, cy + (l % 9) // 3 - 1, cz + l // 9 - 1)
    for p, i, k in fine:
        best = take(best, _d2(p, sel), i, k)
    bound = F(F(0.99) * F(cell3))
    if not (best is not None and best[0] <= F(bound * bound)):
        ux, uy, uz = _cell(sel[0], cellc), _cell(sel[1], cellc), _cell(sel[2], cellc)
        r = 1
        while True:
            limit = min(best[0], F(25)) if best is not None else F(25)
            cand = []
            for dx in range(-r, r + 1):
                for dy in range(-r, r + 1):
                    for dz in range(-r, r + 1):
                        if r > 1 and max(abs(dx), abs(dy), abs(dz)) != r:
                            continue                                          # shells beyond the first block
                        gx, gy, gz = _gap(sel[0], ux + dx, cellc), _gap(sel[1], uy + dy, cellc), _gap(sel[2], uz + dz, cellc)
                        if F(F(F(gx * gx + gy * gy) + gz * gz) * F(0.999)) <= limit:
                            cand += g3c.bucket(ux + dx, uy + dy, uz + dz)
            for p, i, k in cand:
                best = take(best, _d2(p, sel), i, k)
            bc = F(F(F(r) - F(0.01)) * F(cellc)); b2 = F(bc * bc)
            if (best is not None and best[0] <= b2) or b2 >= F(25):
                break
            r += 1
    if best is None or not float(best[0]) < 25.0:
        return None
    closest, cid = best[1], best[2]
    t2 = t3 = None                                                           # (d, seq)
    def consider(p, j, key):
        nonlocal t2, t3
        if j == closest or not (cid - 2 <= key <= cid + 2):
            return
        up = j > closest
        c2 = (key <= cid if up else key >= cid) if plane else (key > cid if up else key < cid)
        c3 = plane and not c2
        d = _d2(p, sel)
        if not float(d) < 25.0:
            return
        seq = (j - closest) if up else 0x40000000 + (closest - j)
        if c2 and (t2 is None or (d, seq) < t2): t2 = (d, seq)
        if c3 and (t3 is None or (d, seq) < t3): t3 = (d, seq)
    done2, done3, lim2, lim3 = False, not plane, F(25), F(25)
    if len(fine) <= sweep_rows * 64:                                         # the block's candidates are still in registers
        for p, i, k in fine:
            consider(p, i, k)
        b2 = F(bound * bound)
        if t2 is not None: lim2 = t2[0]
        if plane and t3 is not None: lim3 = t3[0]
        done2 = t2 is not None and lim2 <= b2
        done3 = (not plane) or (t3 is not None and lim3 <= b2)
    level = 0
    while level < 2 and not (done2 and done3):
        cx2, cy2 = _cell(sel[0], cell2), _cell(sel[1], cell2)
        cells = [(a, b) for b in (-1, 0, 1) for a in (-1, 0, 1)] if level == 0 else \
                [(a, b) for a in range(-2, 3) for b in range(-2, 3) if max(abs(a), abs(b)) == 2]
        for a, b in cells:
            for key in range(cid - 2, cid + 3):
                second = (key == cid) if plane else (key != cid)
                wanted = (not done2) if second else (plane and not done3)
                gx, gy = _gap(sel[0], cx2 + a, cell2), _gap(sel[1], cy2 + b, cell2)
                if key >= 0 and wanted and F(F(gx * gx + gy * gy) * F(0.999)) <= (lim2 if second else lim3):
                    for p, i, k in g2.bucket(cx2 + a, cy2 + b, key):
                        consider(p, i, k)
        bnd = F(F(F(level + 1) - F(0.01)) * F(cell2)); b2 = F(bnd * bnd)
        if b2 >= F(25):
            break
        if t2 is not None: lim2 = t2[0]
        if plane and t3 is not None: lim3 = t3[0]
        done2 = t2 is not None and lim2 <= b2
        done3 = (not plane) or (t3 is not None and lim3 <= b2)
        level += 1
    if t2 is None or (plane and t3 is None):
        return (closest, None, None)
    dec = lambda s: closest - (s - 0x40000000) if s >= 0x40000000 else closest + s
    return (closest, dec(t2[1]), dec(t3[1]) if plane else None)


def reference_query(sel, pts, keys, plane):
    """src/laserOdometry.cpp:299-384 / :387-483 as written (exact 1-NN, lowest index on ties; walk until break)."""
    d = np.array([_d2(p, sel) for p in pts], dtype=np.float32)
    closest = int(np.lexsort((np.arange(len(pts)), d))[0])
    if not float(d[closest]) < 25.0:
        return None
    cid = int(keys[closest]); min2 = min3 = None; m2 = m3 = 25.0
    for j in range(closest + 1, len(pts)):
        if plane:
            if keys[j] > cid + 2.5: break
            dj = float(d[j])
            if keys[j] <= cid and dj < m2: m2, min2 = dj, j
            elif keys[j] > cid and dj < m3: m3, min3 = dj, j
        else:
            if keys[j] <= cid: continue
            if keys[j] > cid + 2.5: break
            if float(d[j]) < m2: m2, min2 = float(d[j]), j
    for j in range(closest - 1, -1, -1):
        if plane:
            if keys[j] < cid - 2.5: break
            dj = float(d[j])
            if keys[j] >= cid and dj < m2: m2, min2 = dj, j
            elif keys[j] < cid and dj < m3: m3, min3 = dj, j
        else:
            if keys[j] >= cid: continue
            if keys[j] < cid - 2.5: break
            if float(d[j]) < m2: m2, min2 = float(d[j]), j
    if min2 is None or (plane and min3 is None):
        return (closest, None, None)
    return (closest, min2, min3 if plane else None)


def _cloud(rng, n, rings, extent, voxel):
    """Ring-sorted cloud (keys ascending with the index) on a few surfaces, coordinates snapped to make exact ties likely."""
    keys = np.sort(rng.integers(0, rings, n))
    pts = np.zeros((n, 3), np.float32)
    pts[:, 0] = rng.uniform(-extent, extent, n); pts[:, 1] = rng.uniform(-extent, extent, n)
    pts[:, 2] = np.where(rng.random(n) < 0.7, -1.7 + 0.02 * rng.normal(size=n), rng.uniform(-1.7, 4.0, n))
    snap = rng.random(n) < 0.3
    pts[snap] = np.round(pts[snap] / voxel) * voxel
    return pts, keys


@pytest.mark.parametrize("plane,cell3,H,seed", [(True, 0.5, 256, 1), (True, 0.5, 4096, 2), (False, 1.0, 128, 3), (False, 1.0, 2048, 4),
                                                (True, 0.5, 64, 5), (False, 1.0, 32, 6),
                                                (False, 0.75, 2048, 7), (False, 0.75, 128, 8), (False, 0.75, 32, 9)])   # round 3: the corner class uses 0.75 m cells (1 / 0.75 is not an exact f32)
def test_grid_search_model_equals_the_walk_until_break_definition(plane, cell3, H, seed):
    rng = np.random.default_rng(seed)
    for trial in range(4):
        n = int(rng.integers(150, 1200))
        extent = float(rng.choice([4.0, 12.0, 40.0]))
        pts, keys = _cloud(rng, n, 16, extent, 0.25)
        g3 = Grid(pts, keys, cell3, H, False)
        g3c = Grid(pts, keys, cell3 * 4.0, H, False)
        g2 = Grid(pts, keys, 2.625, H, True)
        nq = 120
        qs = np.zeros((nq, 3), np.float32)
        near = rng.random(nq) < 0.7                                           # most queries near a cloud point, some anywhere (far tails)
        qs[near] = pts[rng.integers(0, n, near.sum())] + rng.normal(scale=rng.choice([0.0, 0.05, 0.6]), size=(near.sum(), 3)).astype(np.float32)
        qs[~near] = rng.uniform(-1.3 * extent, 1.3 * extent, ((~near).sum(), 3)).astype(np.float32)
        for qi, sel in enumerate(qs):
            got = model_query(sel, g3, g3c, g2, plane, 3 if plane else 2)
            want = reference_query(sel, pts, keys, plane)
            assert got == want, (plane, cell3, H, seed, trial, qi, sel, got, want)


def test_map_search_block_model_equals_brute_force():
    """k_map_search (a-loam_amd/csrc/mapping_kernels.hip): 2 m cells, the 2x2x2 block on the query's side of its cell, duplicate
    buckets walked once, no per-candidate cell test — must find exactly the points with d < 1 m, and from them the five
    smallest (distance, index), which is what the reference uses of nearestKSearch(k = 5) (src/laserMapping.cpp:582,650)."""
    rng = np.random.default_rng(11)
    for H in (8, 64, 4096):
        n = 3000
        pts = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
        snap = rng.random(n) < 0.3
        pts[snap] = np.round(pts[snap] * 2) / 2                                # many points on cell borders / exact ties
        buckets = [[] for _ in range(H)]
        for i, p in enumerate(pts):
            buckets[_hash3(int(np.floor(F(p[0]) * F(0.5))), int(np.floor(F(p[1]) * F(0.5))), int(np.floor(F(p[2]) * F(0.5))), H)].append(i)
        qs = np.concatenate([pts[rng.integers(0, n, 150)] + rng.normal(scale=0.3, size=(150, 3)).astype(np.float32),
                             np.round(rng.uniform(-9, 9, (50, 3)) * 2).astype(np.float32) / 2])
        for sel in qs.astype(np.float32):
            g = [F(sel[k]) * F(0.5) for k in range(3)]
            c = [int(np.floor(v)) for v in g]
            nb = [c[k] + 1 if F(g[k] - F(c[k])) >= F(0.5) else c[k] - 1 for k in range(3)]
            hs = []
            for m in range(8):
                h = _hash3(nb[0] if m & 1 else c[0], nb[1] if m & 2 else c[1], nb[2] if m & 4 else c[2], H)
                if h not in hs:
                    hs.append(h)
            found = sorted((float(_d2(pts[i], sel)), i) for h in hs for i in buckets[h] if _d2(pts[i], sel) < F(1.0))[:5]
            want = sorted((float(_d2(p, sel)), i) for i, p in enumerate(pts) if _d2(p, sel) < F(1.0))[:5]
            assert found == want, (H, sel, found, want)
