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


def _map_bucket(x, y, z, H):
    """map_bucket() of mapping_kernels.hip: parities of the 2 m cell in the low three bits, hash of the 4 m super-cell above them."""
    m = 0xFFFFFFFF
    h = ((((x >> 1) & m) * 73856093 & m) ^ (((y >> 1) & m) * 19349663 & m) ^ (((z >> 1) & m) * 83492791 & m))
    return ((h << 3 & m) | (x & 1) | ((y & 1) << 1) | ((z & 1) << 2)) & (H - 1)


def _map_block(sel):
    g = [F(sel[k]) * F(0.5) for k in range(3)]
    c = [int(np.floor(v)) for v in g]
    nb = [c[k] + 1 if F(g[k] - F(c[k])) >= F(0.5) else c[k] - 1 for k in range(3)]
    return c, nb


def _keys5_insert(t, k):
    """Keys5::insert: branch-free insertion of key k into the ascending five."""
    c = [k < x for x in t]
    return [k if c[0] else t[0],
            t[0] if c[0] else (k if c[1] else t[1]),
            t[1] if c[1] else (k if c[2] else t[2]),
            t[2] if c[2] else (k if c[3] else t[3]),
            t[3] if c[3] else (k if c[4] else t[4])]


def _bits(d):
    return int(np.array([d], np.float32).view(np.uint32)[0])


def test_map_search_block_model_equals_brute_force():
    """k_map_search, round 5 (a-loam_amd/csrc/mapping_kernels.hip): 2 m cells whose buckets carry the cell parities in their low
    bits (the eight cells of the 2x2x2 block on the query's side never share a bucket: no duplicate test), no per-candidate cell test,
    the five smallest kept as packed (distance bits << 32 | index) keys by a branch-free insertion with the entry's position beside
    them — must give exactly the five smallest (distance, index) among the points with d < 1 m, or nothing when there are fewer than
    five, which is what the reference uses of nearestKSearch(k = 5) (src/laserMapping.cpp:582,650)."""
    rng = np.random.default_rng(11)
    for H in (8, 64, 4096):
        n = 3000
        pts = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
        snap = rng.random(n) < 0.3
        pts[snap] = np.round(pts[snap] * 2) / 2                                # many points on cell borders / exact ties
        buckets = [[] for _ in range(H)]
        for i, p in enumerate(pts):
            buckets[_map_bucket(int(np.floor(F(p[0]) * F(0.5))), int(np.floor(F(p[1]) * F(0.5))), int(np.floor(F(p[2]) * F(0.5))), H)].append(i)
        for bk in buckets:
            rng.shuffle(bk)                                                    # the LDS counting sort fills a bucket in no particular order
        qs = np.concatenate([pts[rng.integers(0, n, 150)] + rng.normal(scale=0.3, size=(150, 3)).astype(np.float32),
                             np.round(rng.uniform(-9, 9, (50, 3)) * 2).astype(np.float32) / 2]).astype(np.float32)
        for sel in qs:
            c, nb = _map_block(sel)
            hs = [_map_bucket(nb[0] if m & 1 else c[0], nb[1] if m & 2 else c[1], nb[2] if m & 4 else c[2], H) for m in range(8)]
            assert len(set(hs)) == 8, "the eight cells of a block must sit in eight different buckets"
            t, pos = [0x3F800000 << 32] * 5, [None] * 5
            for h in hs:
                for i in buckets[h]:
                    d = _d2(pts[i], sel)
                    if d < F(1.0):                                             # the kernel's wave-level skip; the sentinel alone would do
                        key = _bits(d) << 32 | i
                        cmp = [key < x for x in t]
                        t = _keys5_insert(t, key)
                        pos = [i if cmp[0] else pos[0]] + [pos[k - 1] if cmp[k - 1] else (i if cmp[k] else pos[k]) for k in range(1, 5)]
            found = pos if (t[4] >> 32) < 0x3F800000 else None
            w = sorted((float(_d2(p, sel)), i) for i, p in enumerate(pts) if _d2(p, sel) < F(1.0))[:5]
            want = [i for _, i in w] if len(w) == 5 else None
            assert found == want, (H, sel, found, want)


def test_keys5_insertion_is_a_sorted_top_five():
    rng = np.random.default_rng(5)
    for _ in range(300):
        m = int(rng.integers(0, 12))
        keys = [int(k) for k in rng.integers(0, 1 << 62, m)]
        keys += keys[:2]                                                       # equal keys cannot occur (index and slot differ); harmless if they did
        t = [1 << 62] * 5
        for k in keys:
            t = _keys5_insert(t, k)
        want = sorted(keys + [1 << 62] * 5)[:5]
        assert t == want


# ======================================================================================================================
# Round 4: k_associate_pair.  (a) the owner of a list position from bit masks + v_mbcnt (sweep2), (b) the search structure
# of a query: fine block with kept keys and the per-query radius, lanes past the list reading ARBITRARY real entries,
# classes by "same ring / other rings", tails in stages (own ring-grid cell first), the axis_gap() form of the cell
# distances — against the walk-until-break definition above.
# ======================================================================================================================
def _mbcnt(mask_lo, mask_hi, lane, add):
    """v_mbcnt_hi(mask_hi, v_mbcnt_lo(mask_lo, add)) for one lane."""
    lo = bin(mask_lo & ((1 << min(lane, 32)) - 1)).count("1")
    hi = bin(mask_hi & ((1 << max(lane - 32, 0)) - 1)).count("1")
    return add + lo + hi


def model_sweep2(s0, cnt, halves, k_rows, n_entries, rng):
    """What every lane of every row reads, as the kernel computes it: list of rounds, each [row][lane] -> entry index.  s0 / cnt: per-lane
    bucket bounds (64).  Positions past the end of a list get an entry index from a garbage table slot, clamped into the cloud."""
    W = 32 if halves else 64
    segs = [range(0, 32), range(32, 64)] if halves else [range(0, 64)]
    incl = [0] * 64
    for seg in segs:
        run = 0
        for l in seg:
            run += cnt[l]; incl[l] = run
    total = [incl[31], incl[63]] if halves else [incl[63]]
    tmax = max(total)
    rounds = []
    if tmax <= 0:
        return rounds, total
    nonempty = [c > 0 for c in cnt]
    table = [int(rng.integers(-5 * n_entries, 5 * n_entries)) for _ in range(72)]      # stale slots hold anything
    rank_own = [sum(nonempty[:l]) for l in range(64)]
    for l in range(64):
        if nonempty[l]:
            table[rank_own[l]] = s0[l] - (incl[l] - cnt[l])
    rank0 = [0, sum(nonempty[:32])] if halves else [0]
    base = 0
    while base < tmax:
        rows = k_rows if tmax - base > (k_rows - 1) * W else (tmax - base + W - 1) // W
        marks = [[0, 0, 0] for _ in range(k_rows * (2 if halves else 1))]                 # slot: lo, hi, both
        for l in range(64):
            pr = incl[l] - 1 - base
            if nonempty[l] and 0 <= pr < k_rows * W:
                r, bit, h = pr // W, pr % W, (l >> 5 if halves else 0)
                slot = marks[r * 2 + h] if halves else marks[r]
                if halves:
                    slot[h] |= 1 << bit; slot[2] |= 1 << bit
                else:
                    slot[bit >> 5] |= 1 << (bit & 31)
        out = []
        carry = []
        for h in range(len(segs)):
            before = sum(1 for l in segs[h] if nonempty[l] and incl[l] - 1 < base) if base > 0 else 0
            carry.append(rank0[h] + before)
        for u in range(rows):
            row = []
            for lane in range(64):
                h = lane >> 5 if halves else 0
                m = marks[u * 2 + h] if halves else marks[u]
                rank = _mbcnt(m[0], m[1], lane, carry[h])
                at = table[rank] + base + u * W + (lane & (W - 1))
                at &= 0xFFFFFFFF                                                          # unsigned compare
                row.append(at if at < n_entries - 1 else n_entries - 1)
            for h in range(len(segs)):
                m = marks[u * 2 + h] if halves else marks[u]
                carry[h] += bin(m[2]).count("1") if halves else bin(m[0]).count("1") + bin(m[1]).count("1")
            out.append(row)
        rounds.append(out)
        base += k_rows * W
    return rounds, total


@pytest.mark.parametrize("halves,k_rows", [(True, 6), (True, 4), (False, 3), (False, 2), (True, 1)])
def test_sweep2_owner_masks_deal_every_list_position_to_its_entry(halves, k_rows):
    """Bit mask at the last position of every non-empty bucket + v_mbcnt + running popcount = rank of the owner; the table at that rank turns
    a position into its entry.  Every position of every list must get exactly the entry a plain concatenation of the buckets has there,
    over several rounds, with empty buckets anywhere, empty halves and buckets that end exactly at row / round borders; every other lane
    reads some entry inside the cloud."""
    rng = np.random.default_rng(7)
    W = 32 if halves else 64
    for trial in range(400):
        n_entries = int(rng.integers(50, 5000))
        nb = int(rng.integers(0, 33 if halves else 65))
        cnt, s0 = [0] * 64, [0] * 64
        for seg in ([range(0, 32), range(32, 64)] if halves else [range(0, 64)]):
            lanes = sorted(rng.choice(list(seg), size=min(nb, len(seg)), replace=False)) if rng.random() > 0.1 else []
            for l in lanes:
                style = rng.random()
                c = int(rng.integers(0, 4)) if style < 0.3 else int(rng.integers(0, 40)) if style < 0.9 else W * int(rng.integers(1, 3))
                c = min(c, n_entries)
                cnt[l], s0[l] = c, int(rng.integers(0, n_entries - c + 1))
        rounds, total = model_sweep2(s0, cnt, halves, k_rows, n_entries, rng)
        want = []
        for seg in ([range(0, 32), range(32, 64)] if halves else [range(0, 64)]):
            want.append([s0[l] + k for l in seg for k in range(cnt[l])])
        got = [[] for _ in want]
        for out in rounds:
            for row in out:
                for h in range(len(want)):
                    lanes = range(h * 32, h * 32 + 32) if halves else range(64)
                    got[h] += [row[l] for l in lanes]
                assert all(0 <= a < n_entries for a in row)
        for h in range(len(want)):
            assert total[h] == len(want[h])
            assert got[h][:len(want[h])] == want[h], (trial, h)


class SortedGrid(Grid):
    """The same buckets laid out like the counting sort leaves them (bucket after bucket), so that "the entry at index k" exists."""

    def __init__(self, pts, keys, cell, H, third_is_key):
        super().__init__(pts, keys, cell, H, third_is_key)
        self.flat = [e for b in self.b for e in b]


def _axis_gap(d, ad_cell, s, c, cell):
    lo = F(F(c) * F(cell))
    up = F(F(F(F(lo + F(cell)) - F(s)) - F(cell)) - F(1e-3)); dn = F(F(F(F(s) - lo) - F(cell)) - F(1e-3))
    return max(F(F(ad_cell) + (up if d > 0 else dn)), F(0))


def model_query_pair(sel, g3, g3c, g2, plane, k_rows, rng, own_first):
    """One query through the round-4 structure.  Lanes past the end of a candidate list look at random real entries of the same grid."""
    cell3, cellc, cell2 = g3.cell, g3c.cell, g2.cell
    def extras(g, n_list, width):                                            # what the idle lanes of the rows in use read
        pad = (-n_list) % width
        return [g.flat[int(rng.integers(0, len(g.flat)))] for _ in range(pad)] if g.flat else []
    best = None
    def take(cur, d, idx, key):
        return (d, idx, key) if cur is None or (d, idx) < (cur[0], cur[1]) else cur
    cx, cy, cz = _cell(sel[0], cell3), _cell(sel[1], cell3), _cell(sel[2], cell3)
    fine = []
    for l in range(27):
        fine += g3.bucket(cx + l % 3 - 1, cy + (l % 9) // 3 - 1, cz + l // 9 - 1)
    single_round = len(fine) <= k_rows * 32
    kept = (fine + extras(g3, len(fine), 32)) if single_round else fine[:k_rows * 32]
    for p, i, k in fine + extras(g3, len(fine), 32):
        best = take(best, _d2(p, sel), i, k)
    # per-query radius of the block: one cell + the distance to the nearest face of the own cell, minus 1 mm, times 0.99
    inv = F(1.0) / F(cell3)
    lows = [F(np.floor(F(s) * inv) * F(cell3)) for s in sel]
    gap = min(min(F(F(s) - lo), F(F(lo + F(cell3)) - F(s))) for s, lo in zip(sel, lows))
    b = F(F(0.99) * F(F(cell3) + (F(gap - F(1e-3)) if gap > F(1e-3) else F(0))))
    fine_b2 = F(b * b)
    if not (best is not None and best[0] <= fine_b2):
        ux, uy, uz = _cell(sel[0], cellc), _cell(sel[1], cellc), _cell(sel[2], cellc)
        r = 1
        while True:
            limit = min(best[0], F(25)) if best is not None else F(25)
            cand = []
            for dx in range(-r, r + 1):
                for dy in range(-r, r + 1):
                    for dz in range(-r, r + 1):
                        if r > 1 and max(abs(dx), abs(dy), abs(dz)) != r:
                            continue
                        gx = _axis_gap(dx, F(abs(dx)) * F(cellc), sel[0], ux, cellc); gy = _axis_gap(dy, F(abs(dy)) * F(cellc), sel[1], uy, cellc)
                        gz = _axis_gap(dz, F(abs(dz)) * F(cellc), sel[2], uz, cellc)
                        if F(F(F(gx * gx + gy * gy) + gz * gz) * F(0.999)) <= limit:
                            cand += g3c.bucket(ux + dx, uy + dy, uz + dz)
            for p, i, k in cand + extras(g3c, len(cand), 64):
                best = take(best, _d2(p, sel), i, k)
            bc = F(F(F(r) - F(0.01)) * F(cellc)); b2 = F(bc * bc)
            if (best is not None and best[0] <= b2) or b2 >= F(25):
                break
            r += 1
    if best is None or not float(best[0]) < 25.0:
        return None
    closest, cid = best[1], best[2]
    t2 = t3 = None                                                           # (d, seq) with the round-4 order key
    def consider(p, j, key):
        nonlocal t2, t3
        d = _d2(p, sel)
        t = j - closest
        seq = (max(t, -t) | (0x80000000 if t < 0 else 0)) & 0xFFFFFFFF
        ok = j != closest and 0 <= key - cid + 2 <= 4 and float(d) < 25.0
        own = key == cid
        if plane:
            if ok and own and (t2 is None or (d, seq) < t2): t2 = (d, seq)
            if ok and not own and (t3 is None or (d, seq) < t3): t3 = (d, seq)
        elif ok and not own and (t2 is None or (d, seq) < t2): t2 = (d, seq)
    want2, want3 = True, plane
    if single_round:
        for p, i, k in kept:
            consider(p, i, k)
        if t2 is not None and t2[0] <= fine_b2: want2 = False
        if plane and t3 is not None and t3[0] <= fine_b2: want3 = False
    lim2 = t2[0] if t2 is not None else F(25); lim3 = t3[0] if t3 is not None else F(25)
    cx2, cy2 = _cell(sel[0], cell2), _cell(sel[1], cell2)
    ring16 = [(a, c) for a in range(-2, 3) for c in range(-2, 3) if max(abs(a), abs(c)) == 2]
    stage = 0 if own_first else 1
    while stage < 3 and (want2 or want3):
        cells = [(0, 0)] if stage == 0 else ([(a, c) for c in (-1, 0, 1) for a in (-1, 0, 1) if own_first is False or (a, c) != (0, 0)] if stage == 1 else ring16)
        wo, ws = (want3 if plane else want2), (plane and want2)
        cand = []
        for a, c in cells:
            for slot in range(5 if plane else 4):
                other = slot < 4
                if not (wo if other else ws):
                    continue
                key = cid - 2 + ((0x2819 >> (slot * 3)) & 7)
                gx = _axis_gap(a, F(abs(a)) * F(cell2), sel[0], cx2, cell2) if (a, c) != (0, 0) else F(0)
                gy = _axis_gap(c, F(abs(c)) * F(cell2), sel[1], cy2, cell2) if (a, c) != (0, 0) else F(0)
                second = (not other) if plane else True
                if key >= 0 and F(F(gx * gx + gy * gy) * F(0.999)) <= (lim2 if second else lim3):
                    cand += g2.bucket(cx2 + a, cy2 + c, key)
        for p, i, k in cand + extras(g2, len(cand), 64):
            consider(p, i, k)
        if stage == 2:
            break
        if stage == 0:
            lox, loy = F(F(cx2) * F(cell2)), F(F(cy2) * F(cell2))
            ups = [F(F(F(F(lo + F(cell2)) - F(s)) - F(cell2)) - F(1e-3)) for s, lo in ((sel[0], lox), (sel[1], loy))]
            dns = [F(F(F(F(s) - lo) - F(cell2)) - F(1e-3)) for s, lo in ((sel[0], lox), (sel[1], loy))]
            bound = F(F(0.99) * max(F(min(min(ups[0], dns[0]), min(ups[1], dns[1])) + F(cell2)), F(0)))
        else:
            bound = F(F(F(stage) - F(0.01)) * F(cell2))
        b2 = F(bound * bound)
        if t2 is not None: lim2 = t2[0]
        if plane and t3 is not None: lim3 = t3[0]
        if t2 is not None and lim2 <= b2: want2 = False
        if plane and t3 is not None and lim3 <= b2: want3 = False
        stage += 1
    if t2 is None or (plane and t3 is None):
        return (closest, None, None)
    dec = lambda s: closest - (s & 0x7FFFFFFF) if s & 0x80000000 else closest + s
    return (closest, dec(t2[1]), dec(t3[1]) if plane else None)


@pytest.mark.parametrize("plane,cell3,H,seed,own_first", [(True, 0.5, 256, 21, True), (True, 0.5, 4096, 22, True), (True, 0.5, 64, 23, True), (True, 0.5, 512, 24, False),
                                                          (False, 0.75, 2048, 25, False), (False, 0.75, 128, 26, False), (False, 0.75, 32, 27, False), (False, 0.75, 256, 28, True)])
def test_pair_kernel_search_model_equals_the_walk_until_break_definition(plane, cell3, H, seed, own_first):
    """The structure of k_associate_pair for one query — per-query block radius, kept keys, idle lanes looking at random real entries, classes
    as "same ring" / "other rings" on a ring-sorted cloud, the (|j - closest|, direction) order key, ring-grid stages with the own cell
    first and its face-distance bound, axis_gap() — gives exactly what the reference's literal loops give, on clouds with collisions,
    snapped coordinates (exact ties) and far queries."""
    rng = np.random.default_rng(seed)
    for trial in range(4):
        n = int(rng.integers(150, 1200))
        extent = float(rng.choice([4.0, 12.0, 40.0]))
        pts, keys = _cloud(rng, n, 16, extent, 0.25)
        g3 = SortedGrid(pts, keys, cell3, H, False)
        g3c = SortedGrid(pts, keys, cell3 * 4.0, H, False)
        g2 = SortedGrid(pts, keys, 2.625, H, True)
        nq = 120
        qs = np.zeros((nq, 3), np.float32)
        near = rng.random(nq) < 0.7
        qs[near] = pts[rng.integers(0, n, near.sum())] + rng.normal(scale=rng.choice([0.0, 0.05, 0.6]), size=(near.sum(), 3)).astype(np.float32)
        qs[~near] = rng.uniform(-1.3 * extent, 1.3 * extent, ((~near).sum(), 3)).astype(np.float32)
        for qi, sel in enumerate(qs):
            got = model_query_pair(sel, g3, g3c, g2, plane, 6 if plane else 4, rng, own_first)
            want = reference_query(sel, pts, keys, plane)
            assert got == want, (plane, cell3, H, seed, trial, qi, sel, got, want)


# ---- nearly ring-sorted clouds (round 6): the index-range form of the walk window (odometry_kernels.hip walk_tables / consider2<.., true>) ----------------
def _literal_walks(keys, d, closest, plane):
    """reference src/laserOdometry.cpp:315-361 (corner) / :410-455 (plane), on a cloud whose ring ids are `keys` and whose squared distances to the query are `d`."""
    c, n = keys[closest], len(keys)
    if not plane:
        m, ind = 25.0, -1
        for j in range(closest + 1, n):
            if keys[j] <= c:
                continue
            if keys[j] > c + 2.5:
                break
            if d[j] < m:
                m, ind = d[j], j
        for j in range(closest - 1, -1, -1):
            if keys[j] >= c:
                continue
            if keys[j] < c - 2.5:
                break
            if d[j] < m:
                m, ind = d[j], j
        return ind, -1
    m2 = m3 = 25.0
    i2 = i3 = -1
    for j in range(closest + 1, n):
        if keys[j] > c + 2.5:
            break
        if keys[j] <= c and d[j] < m2:
            m2, i2 = d[j], j
        elif keys[j] > c and d[j] < m3:
            m3, i3 = d[j], j
    for j in range(closest - 1, -1, -1):
        if keys[j] < c - 2.5:
            break
        if keys[j] >= c and d[j] < m2:
            m2, i2 = d[j], j
        elif keys[j] < c and d[j] < m3:
            m3, i3 = d[j], j
    return i2, i3


def _range_form(keys, d, closest, plane, R):
    """What k_associate_nearly computes: candidates = indices inside (last[c - 3], first[c + 3]), class by direction and key, minimum of (distance, visit order)."""
    n, S, c = len(keys), R + 8, keys[closest]
    first = [next((i for i in range(n) if keys[i] >= k), n) for k in range(S)]
    last = [next((i for i in range(n - 1, -1, -1) if keys[i] <= k), -1) for k in range(S)]
    assert all(last[k - 3] < first[k] for k in range(3, S)), "not a nearly ring-sorted cloud (flags[1] would be 2)"
    hi, lo = first[c + 3], (last[c - 3] if c >= 3 else -1)
    best = {2: (np.inf, 0, -1), 3: (np.inf, 0, -1)}
    for j in range(n):
        if j == closest or not (lo < j < hi) or not d[j] < 25.0:
            continue
        t = j - closest
        seq = t if t > 0 else (1 << 31) - t
        own = keys[j] <= c if t > 0 else keys[j] >= c
        cls = (2 if own else 3) if plane else (None if own else 2)
        if cls is not None and (d[j], seq) < best[cls][:2]:
            best[cls] = (d[j], seq, j)
    return best[2][2], best[3][2]


def test_index_range_form_of_the_walk_window_on_nearly_ring_sorted_clouds():
    """A sweep whose first ray had no return carries ring ids one too low in the middle of every ring's stretch (negative relTime, reference
    src/scanRegistration.cpp:211-214,239).  As long as no id lies more than 2 below an earlier one, the reference's walks visit exactly the index range
    (last index with id <= c - 3, first index with id >= c + 3), and the neighbours are the (distance, visit order) minima of that range under the
    direction-dependent class rules - including exact distance ties, ids two too low, empty rings and queries at either end of the cloud."""
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(2500):
        R = int(rng.integers(6, 20))
        keys = []
        for r in range(R):
            for _k in range(int(rng.integers(0, 12))):
                keys.append(max(0, r - (int(rng.integers(1, 3)) if rng.random() < 0.25 else 0)))
        keys = np.array(keys, int)
        if len(keys) < 3 or (np.maximum.accumulate(keys) - keys).max() > 2:
            continue
        d = rng.choice([1.0, 2.0, 3.0, 4.0, 30.0, 0.5], size=len(keys)) + rng.integers(0, 3, len(keys))      # few distinct values: ties everywhere
        closest = int(rng.integers(0, len(keys)))
        for plane in (False, True):
            assert _literal_walks(keys, d, closest, plane) == _range_form(keys, d, closest, plane, R), (keys.tolist(), d.tolist(), closest, plane)
            checked += 1
    assert checked > 3000
