"""Model check of the corner / flat selection the HIP kernel performs (a-loam_amd/csrc/registration_kernels.hip,
pick_sector + the second pass of k_ring_features).

The reference sorts every sector by curvature and walks it, skipping points already marked by earlier picks — marks that
spill over sector borders, so its six sectors are inherently sequential (src/scanRegistration.cpp:284-390).  The kernel
runs the six sectors concurrently WITHOUT the incoming marks (iterative arg-max / arg-min instead of a sort) and afterwards
redoes, in order, only the sectors that picked a point the sectors before them had marked.  This file runs that scheme as a
Python model against the literal sequential definition over thousands of random rings — ties in curvature, gaps that cut the
neighbour suppression short, sectors shorter than the suppression reach — so the soundness of the speculation does not rest
on the handful of sweeps the GPU tests see."""
import numpy as np
import pytest

F = np.float32


def literal(curv, gap, n):
    """-> (labels, per-sector pick lists) by the reference's sequential walk.  gap[i]: the step i -> i+1 is longer than the
    0.05 threshold.  Ties: ascending (curvature, index), the canonical order (oracle default and the HIP path)."""
    L = n - 11
    picked = np.zeros(n, bool); label = np.zeros(n, np.int8); out = []
    for j in range(6):
        sp, ep = 5 + (L * j) // 6, 5 + (L * (j + 1)) // 6 - 1
        order = sorted(range(sp, ep + 1), key=lambda i: (curv[i], i))
        corners, flats = [], []
        def mark(ind):
            picked[ind] = True
            for l in range(1, 6):
                if gap[ind + l - 1]: break
                picked[ind + l] = True
            for l in range(-1, -6, -1):
                if gap[ind + l]: break
                picked[ind + l] = True
        cnt = 0
        for ind in reversed(order):
            if not picked[ind] and float(curv[ind]) > 0.1:
                cnt += 1
                if cnt <= 2: label[ind] = 2
                elif cnt <= 20: label[ind] = 1
                else: break
                corners.append(ind)
                mark(ind)
        cnt = 0
        for ind in order:
            if not picked[ind] and float(curv[ind]) < 0.1:
                label[ind] = -1
                flats.append(ind)
                cnt += 1
                if cnt >= 4: break
                mark(ind)
        out.append((corners, flats))
    return label, out


def pick_sector(j, init_marks, L, curv, reach):
    """One sector on its own: iterative arg-max / arg-min over the still-unpicked points.  -> corners, flats, spill mask."""
    sp, ln = (L * j) // 6, (L * (j + 1)) // 6 - (L * j) // 6
    first, last = sp + 5, sp + ln - 1 + 5
    alive = [not (p < 5 and (init_marks >> p) & 1) for p in range(ln)]
    spill = 0
    def pick_at(pos):
        nonlocal spill
        fw, bk = reach[first + pos]
        for q in range(max(0, pos - bk), min(ln - 1, pos + fw) + 1):
            alive[q] = False
        kf = first + pos
        for off in range(1, fw + 1):
            if kf + off > last:
                spill |= 1 << (kf + off - last - 1)
        return kf
    corners, flats = [], []
    count = 0
    while True:
        cand = [(curv[first + p], p) for p in range(ln) if alive[p] and curv[first + p] != 0]
        if not cand: break
        c, p = max(cand)                                                     # ties: the larger index
        if not float(c) > 0.1: break
        count += 1
        if count > 20: break
        corners.append(pick_at(p))
    count = 0
    while True:
        cand = [(curv[first + p], p) for p in range(ln) if alive[p]]
        if not cand: break
        c, p = min(cand)                                                     # ties: the smaller index
        if not float(c) < 0.1: break
        flats.append(first + p)
        count += 1
        if count >= 4: break
        pick_at(p)
    return corners, flats, spill


def model(curv, gap, n):
    L = n - 11
    reach = []
    for i in range(n):
        fw = bk = 0
        while fw < 5 and i + fw < n - 1 and not gap[i + fw]: fw += 1
        while bk < 5 and i - 1 - bk >= 0 and not gap[i - 1 - bk]: bk += 1
        reach.append((fw, bk))
    res = [pick_sector(j, 0, L, curv, reach) for j in range(6)]              # concurrently, no incoming marks
    carry = res[0][2]
    for j in range(1, 6):
        sp, ln = (L * j) // 6, (L * (j + 1)) // 6 - (L * j) // 6
        m = carry if ln >= 5 else carry & ((1 << ln) - 1)
        if m:
            hit = any(pk - 5 - sp < 5 and (m >> (pk - 5 - sp)) & 1 for pk in res[j][0] + res[j][1])
            if hit:
                res[j] = pick_sector(j, m, L, curv, reach)
        carry = (0 if ln >= 5 else carry >> ln) | res[j][2]
    label = np.zeros(n, np.int8)
    for corners, flats, _ in res:
        for q, ind in enumerate(corners): label[ind] = 2 if q < 2 else 1
        for ind in flats: label[ind] = -1
    return label, [(c, f) for c, f, _ in res]


@pytest.mark.parametrize("seed", range(8))
def test_speculative_sector_selection_equals_the_sequential_walk(seed):
    rng = np.random.default_rng(100 + seed)
    for trial in range(250):
        n = int(rng.choice([17, 18, 20, 23, 29, 35, 41, 47, 60, 90, 150, 400]))
        style = trial % 4
        if style == 0:   curv = rng.exponential(0.2, n)
        elif style == 1: curv = rng.choice([0.0, 0.05, 0.1, 0.2, 1.0], n)                   # massive ties, values on the thresholds
        elif style == 2: curv = np.where(rng.random(n) < 0.5, rng.uniform(0.11, 3, n), rng.uniform(0, 0.09, n))
        else:            curv = np.round(rng.exponential(0.3, n), 1)
        curv = curv.astype(np.float32)
        gap = rng.random(n) < rng.choice([0.0, 0.05, 0.3, 0.8])
        la, pa = literal(curv, gap, n)
        lb, pb = model(curv, gap, n)
        assert np.array_equal(la, lb) and pa == pb, (seed, trial, n, curv.tolist(), gap.tolist(), pa, pb)
