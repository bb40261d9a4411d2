"""CPU models of the arguments the round-2 registration kernels rest on (no GPU, no oracle):

* k_front's ring id: an f32 evaluation of the elevation angle decides the ring wherever the decision is the same 2e-4 degrees
  below and above it; the f64 expression of reference src/scanRegistration.cpp:166 decides elsewhere.  Claim: identical rings.
* k_ring_features' voxel indices from packed integer cells == pcl::VoxelGrid's float arithmetic (floor(p * inv) - floor(min * inv)).
* k_front's offsets inside a block: ranking four rounds first and scanning the 16 (round, wave) counts == the sequential stable compaction.
* k_ring_features' final-place output: offsets gathered from the published ring counts == plain concatenation ring by ring.
"""
import numpy as np

f32 = np.float32


def ring_from_angle(angle, R):
    """reference src/scanRegistration.cpp:169-205 on an f32 angle (same expressions as ring_from_angle in registration_kernels.hip)."""
    angle = f32(angle)
    if R == 16:
        sid = int(np.float64(f32(f32(angle + f32(15)) / f32(2))) + 0.5)
        return -1 if sid > R - 1 or sid < 0 else sid
    if R == 32:
        sid = int((np.float64(angle) + 92.0 / 3.0) * 3.0 / 4.0)
        return -1 if sid > R - 1 or sid < 0 else sid
    if np.float64(angle) >= -8.83:
        sid = int(np.float64(f32(f32(2) - angle)) * 3.0 + 0.5)
    else:
        sid = R // 2 + int((-8.83 - np.float64(angle)) * 2.0 + 0.5)
    if np.float64(angle) > 2 or np.float64(angle) < -24.33 or sid > 50 or sid < 0:
        return -1
    return sid


POLY = [1.0, -0.3333319425582886, 0.19994600117206573, -0.1420508623123169, 0.10522426664829254, -0.06763934344053268, 0.025188861414790154]


def fast_angle(x, y, z, rsq_ulp=0):
    """the f32 fast path: t = z * rsq(x^2 + y^2) (rsq within 1 ulp), degree-6 polynomial of atan(t) / t in t^2, degrees."""
    s2 = f32(f32(x * x) + f32(y * y))
    rs = f32(1.0 / np.sqrt(np.float64(s2)))
    rs = np.nextafter(rs, f32(np.inf) if rsq_ulp > 0 else f32(-np.inf), dtype=f32) if rsq_ulp else rs
    t = f32(z * rs)
    u = f32(t * t)
    acc = f32(POLY[-1])
    for c in POLY[-2::-1]:
        acc = f32(f32(acc * u) + f32(c))
    return f32(f32(t * acc) * f32(57.29577951)), t


def exact_angle(x, y, z):
    return f32(np.arctan(np.float64(z) / np.sqrt(np.float64(f32(f32(x * x) + f32(y * y))))) * 180 / np.pi)


def test_fast_ring_id_equals_the_f64_expression():
    rng = np.random.default_rng(7)
    n = 60000
    # elevations dense around every ring boundary of the three sensors and uniform in between, ranges 0.5 .. 120 m
    el = np.concatenate([rng.uniform(-32, 32, n // 2), (rng.integers(-100, 100, n // 2) / 6.0 + rng.normal(0, 3e-4, n // 2))])
    rg = rng.uniform(0.5, 120.0, n)
    az = rng.uniform(-np.pi, np.pi, n)
    x = (rg * np.cos(np.deg2rad(el)) * np.cos(az)).astype(f32); y = (rg * np.cos(np.deg2rad(el)) * np.sin(az)).astype(f32)
    z = (rg * np.sin(np.deg2rad(el))).astype(f32)
    worst, slow = 0.0, 0
    for i in range(n):
        ex = exact_angle(x[i], y[i], z[i])
        for ulp in (-1, 0, 1):
            fa, t = fast_angle(x[i], y[i], z[i], ulp)
            worst = max(worst, abs(float(fa) - float(ex)))
            for R in (16, 32, 64):
                lo, hi = ring_from_angle(f32(fa - f32(2e-4)), R), ring_from_angle(f32(fa + f32(2e-4)), R)
                if lo == hi and abs(float(t)) <= 0.65:
                    assert lo == ring_from_angle(ex, R), (i, R, float(fa), float(ex))
                elif ulp == 0 and R == 64:
                    slow += 1
    assert worst < 2e-5, worst                        # what the 2e-4 margin has to cover ten times over
    assert slow < 0.6 * n                             # half of this sample sits on boundaries on purpose; uniform data: ~0.1 %


def test_ring_decision_is_monotone_in_the_angle():
    a = np.arange(-32.0, 4.0, 1e-3, dtype=np.float64).astype(f32)
    for R in (16, 32, 64):
        ids = np.array([ring_from_angle(v, R) for v in a])
        valid = ids >= 0
        idx = np.nonzero(valid)[0]
        assert idx.size and np.all(np.diff(idx) == 1)                      # one contiguous accepted interval
        d = np.diff(ids[valid])
        assert np.all(d >= 0) or np.all(d <= 0)                              # monotone inside it


def test_packed_cells_give_pcl_voxel_indices():
    rng = np.random.default_rng(3)
    inv = f32(1.0) / f32(0.2)
    for trial in range(200):
        n = int(rng.integers(20, 600))
        c = rng.uniform(-150, 150, 3)
        pts = (c + rng.normal(0, rng.uniform(0.05, 30), (n, 3))).astype(f32)
        pts[:, 2] = np.clip(pts[:, 2], -90, 90)
        if trial % 7 == 0:
            pts[: n // 3] = np.round(pts[: n // 3] * 5) / 5                  # points exactly on cell borders
        member = rng.random(n) < 0.9
        fl = np.floor((pts * inv).astype(f32)).astype(np.int64)
        ok = (np.abs(fl[:, 0]) < 1024) & (np.abs(fl[:, 1]) < 1024) & (np.abs(fl[:, 2]) < 512)
        if not ok.all() or not member.any():
            continue
        packed = (fl[:, 0] + 1024) | ((fl[:, 1] + 1024) << 11) | ((fl[:, 2] + 512) << 22)
        cx, cy, cz = packed & 2047, (packed >> 11) & 2047, packed >> 22
        cells = np.stack([cx, cy, cz], 1)[member]
        minc = cells.min(0); divc = cells.max(0) - minc + 1
        vi_int = (cells[:, 0] - minc[0]) + (cells[:, 1] - minc[1]) * divc[0] + (cells[:, 2] - minc[2]) * divc[0] * divc[1]
        # pcl::VoxelGrid: float min / max of the members, then floor(p * inv) - floor(min * inv)
        m = pts[member]
        gmn, gmx = m.min(0), m.max(0)
        minb = np.floor((gmn * inv).astype(f32)).astype(np.int64)
        divb = np.floor((gmx * inv).astype(f32)).astype(np.int64) - minb + 1
        ijk = (np.floor((m * inv).astype(f32)) - minb.astype(f32)).astype(np.int64)
        vi_pcl = ijk[:, 0] + ijk[:, 1] * divb[0] + ijk[:, 2] * divb[0] * divb[1]
        assert np.array_equal(divc, divb) and np.array_equal(vi_int, vi_pcl)
        dx = ((gmx - gmn).astype(f32) * inv).astype(np.int64) + 1           # PCL's own overflow guard never exceeds (div_b + 1) per axis
        assert np.all(dx <= divb + 1)


def test_scatter_offsets_equal_sequential_compaction():
    rng = np.random.default_rng(11)
    for _ in range(50):
        R = int(rng.integers(1, 65))
        ring = rng.integers(-1, R, 1024)                                    # -1 = dropped point
        base = rng.integers(0, 1000, R)
        want = np.full(1024, -1); cur = base.copy()
        for i in range(1024):
            if ring[i] >= 0:
                want[i] = cur[ring[i]]; cur[ring[i]] += 1
        cnt = np.zeros((16, R), int); rank = np.zeros(1024, int)
        for k in range(4):
            for w in range(4):
                lanes = np.arange(64) + w * 64 + k * 256
                for r in range(R):
                    sel = lanes[ring[lanes] == r]
                    cnt[k * 4 + w, r] = sel.size
                    rank[sel] = np.arange(sel.size)                           # rank among the same-ring lanes of the wave
        off = base + np.cumsum(cnt, 0) - cnt                                # exclusive over the 16 (round, wave) slots
        got = np.full(1024, -1)
        for i in range(1024):
            if ring[i] >= 0:
                k, w = i // 256, (i % 256) // 64
                got[i] = off[k * 4 + w, ring[i]] + rank[i]
        assert np.array_equal(want, got)


def test_final_place_offsets_from_published_ring_counts():
    rng = np.random.default_rng(5)
    R = 64
    counts = rng.integers(0, 40, (4, R)); counts[:, 51:] = 0                # the unused HDL-64 rings publish zeros
    epoch = 7
    granules = (epoch << 32) | counts                                       # {launch epoch, count}
    stale = ((epoch - 1) << 32) | rng.integers(0, 99, (4, R))               # what the buffer held from the launch before
    for cls in range(4):
        out = []
        for r in range(R):                                                  # ring r gathers the rings in front of it
            g = granules[cls, :r]
            assert np.all(g >> 32 == epoch) and not np.any(stale[cls, :r] >> 32 == epoch)
            out.append(int((g & 0xffffffff).sum()))
        assert out == list(np.concatenate([[0], np.cumsum(counts[cls])[:-1]]))


# ======================================================================================================================
# Round 5: which ring a k_ring_features workgroup works on.  Rounds 2 - 4 took it from blockIdx.y and were correct only while workgroups start
# in linear order; now it is the next ticket of the sweep, taken when the workgroup starts executing (registration_kernels.hip, "output offsets
# across the rings of a sweep").  A scheduler model: workgroups are STARTED in an arbitrary order onto a limited number of resident slots, a
# resident workgroup publishes its count, then waits until all rings in front of it have published, then retires.  With tickets every order
# terminates and every ring gets the offsets of the plain prefix sum; with blockIdx-derived rings an adversarial order fills all slots with
# waiters and nothing ever retires (the bounded spin of the kernel would turn that into wrong offsets).
# ======================================================================================================================
def _run_rings(start_order, n_sweeps, n_rings, slots, counts, ticketed):
    """-> (offsets[sweep][ring] or None on deadlock).  start_order: workgroup ids (sweep, y) in the order the dispatcher starts them."""
    published = [[None] * n_rings for _ in range(n_sweeps)]
    offsets = [[None] * n_rings for _ in range(n_sweeps)]
    ticket = [0] * n_sweeps
    pending, resident = list(start_order), []                                # resident: (sweep, ring) of running workgroups
    while pending or resident:
        progressed = False
        while pending and len(resident) < slots:                             # the dispatcher fills free slots in ITS order
            b, y = pending.pop(0)
            r = ticket[b] if ticketed else y
            ticket[b] += 1
            published[b][r] = counts[b][r]                                   # selection done -> publish (never waits)
            resident.append((b, r))
            progressed = True
        for wg in list(resident):                                            # gather: needs every ring in front of it
            b, r = wg
            if all(published[b][q] is not None for q in range(r)):
                offsets[b][r] = sum(published[b][q] for q in range(r))
                resident.remove(wg)
                progressed = True
        if not progressed:
            return None                                                      # every slot holds a waiter and nothing can start: deadlock
    return offsets


def test_ring_tickets_make_the_lookback_independent_of_the_dispatch_order():
    rng = np.random.default_rng(12)
    n_sweeps, n_rings = 5, 16
    counts = rng.integers(0, 40, (n_sweeps, n_rings)).tolist()
    want = [[sum(counts[b][:r]) for r in range(n_rings)] for b in range(n_sweeps)]
    ids = [(b, y) for y in range(n_rings) for b in range(n_sweeps)]          # the launch's linear order: ring-major over the sweeps
    for slots in (1, 3, 7, 80):
        for trial in range(40):
            order = list(ids)
            if trial:
                rng.shuffle(order)
            assert _run_rings(order, n_sweeps, n_rings, slots, counts, ticketed=True) == want, (slots, trial)
        assert _run_rings(list(ids), n_sweeps, n_rings, slots, counts, ticketed=False) == want      # in-order dispatch: the old scheme worked
    # the old scheme under an order HIP does not forbid: the last rings first, fewer slots than waiters
    backwards = ids[::-1]
    assert _run_rings(backwards, n_sweeps, n_rings, 7, counts, ticketed=False) is None
    assert _run_rings(backwards, n_sweeps, n_rings, 7, counts, ticketed=True) == want
