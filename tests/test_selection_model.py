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


THR = int(np.float32(0.1).view(np.uint32))           # kCurvThresholdBits
INF = 0x7f800000


def pick_sector_regs(j, init_marks, L, curv, reach, W=64, spare=1):
    """pick_sector as the kernel evaluates it since round 3 (ALOAM_RF_PICK 2): point `pos` of the sector in register pos // W of lane
    pos % W; "dead" folded into the key (corner walk: curvature bits + 1, 0 = dead; flat walk: curvature bits, 0xffffffff = dead); the
    0.1 thresholds as integer compares on the bits; a pick kills through lane ranges of register kr and, when the range leaves
    0 .. W-1, of the register before or behind; the corner walk ends after the 20th pick.  W = 16 or 32 instead of 64 makes the
    register borders frequent."""
    sp, ln = (L * j) // 6, (L * (j + 1)) // 6 - (L * j) // 6
    first = sp + 5
    K = (ln + W - 1) // W + spare
    M32 = 0xffffffff
    cc = np.zeros((K, W), np.int64)
    for pos in range(ln):
        if not (pos < 5 and (init_marks >> pos) & 1):
            cc[pos // W, pos % W] = int(np.float32(curv[first + pos]).view(np.uint32)) + 1
    spill = 0
    lanes = np.arange(W)
    def kill(kr, f, dead):
        nonlocal spill
        fw, bk = reach[first + kr * W + f]
        P, lo, w = kr * W + f, f - bk, fw + bk
        cc[kr, ((lanes - lo) & M32) <= w] = dead
        if (lo & M32) > ((W - 1 - w) & M32):
            if lo < 0 and kr > 0: cc[kr - 1, lanes >= W + lo] = dead
            if lo >= 0 and kr + 1 < K: cc[kr + 1, lanes <= lo + w - W] = dead
        if P + 5 >= ln:
            e = P + fw - (ln - 1)
            if e > 0: spill |= (1 << e) - 1
        return P
    corners, flats = [], []
    for count in range(20):
        m = cc.max(axis=0)
        krl = np.zeros(W, np.int64)
        for r in range(1, K): krl = np.where(cc[r] == m, r, krl)
        cmax = int(m.max())
        if not (THR <= ((cmax - 1) & M32) <= INF): break
        tie = np.flatnonzero(m == cmax)
        f = int(tie[0]); kr = int(krl[f])
        if len(tie) != 1:
            wv = int(np.where(m == cmax, krl * W + lanes, 0).max()); f, kr = wv % W, wv // W
        corners.append(first + kill(kr, f, 0))
    cc = (cc - 1) & M32
    count = 0
    while True:
        m = cc.min(axis=0)
        krl = np.full(W, K - 1, np.int64)
        for r in range(K - 2, -1, -1): krl = np.where(cc[r] == m, r, krl)
        cmin = int(m.min())
        if not cmin < THR: break
        tie = np.flatnonzero(m == cmin)
        f = int(tie[0]); kr = int(krl[f])
        if len(tie) != 1:
            wv = int(np.where(m == cmin, krl * W + lanes, M32).min()); f, kr = wv % W, wv // W
        flats.append(first + kr * W + f)
        count += 1
        if count >= 4: break
        kill(kr, f, M32)
    return corners, flats, spill


def model(curv, gap, n, pick_sector=pick_sector):
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


@pytest.mark.parametrize("W", [16, 32, 64])
def test_register_level_selection_equals_the_sequential_walk(W):
    """The loop the kernel runs (keys with "dead" folded in, integer thresholds, lane-range kills through register kr and its
    neighbours) against the reference's literal walk, with register borders every 16 / 32 / 64 points."""
    rng = np.random.default_rng(700 + W)
    sizes = [17, 23, 41, 90, 150, 400, 700] if W < 64 else [41, 150, 400, 900, 2059]
    for trial in range(120 if W < 64 else 40):
        n = int(rng.choice(sizes))
        style = trial % 4
        if style == 0:   curv = rng.exponential(0.2, n)
        elif style == 1: curv = rng.choice([0.0, 0.05, 0.1, 0.2, 1.0], n)
        elif style == 2: curv = np.where(rng.random(n) < 0.5, rng.uniform(0.11, 3, n), rng.uniform(0, 0.09, n))
        else:            curv = np.round(rng.exponential(0.3, n), 1)
        curv = curv.astype(np.float32)
        gap = rng.random(n) < rng.choice([0.0, 0.05, 0.3, 0.8])
        la, pa = literal(curv, gap, n)
        lb, pb = model(curv, gap, n, lambda *a: pick_sector_regs(*a, W=W, spare=trial % 2))
        assert np.array_equal(la, lb) and pa == pb, (W, trial, n)


def test_curvature_thresholds_as_integer_compares():
    """`(double)c > 0.1` == `THR <= bits(c) <= bits(inf)` and `(double)c < 0.1` == `bits(c) < THR` for every non-negative float."""
    rng = np.random.default_rng(5)
    bits = np.concatenate([np.arange(THR - 4096, THR + 4096), rng.integers(0, 0x7f800000, 200000), [0, 1, INF - 1, INF, INF + 1, 0x7fc00000, 0x7fffffff]]).astype(np.uint32)
    with np.errstate(invalid="ignore"):
        c = bits.view(np.float32).astype(np.float64)
        assert np.array_equal(c > 0.1, (bits >= THR) & (bits <= INF))
        assert np.array_equal(c < 0.1, bits < THR)


def test_reach_from_the_gap_bit_window():
    """k_ring_features keeps "step s -> s + 1 is longer than the threshold" as one bit per step (bit s + 64 of an array whose first
    word and whose steps >= n - 1 read 1) and takes the reach of the neighbour suppression around point i from the 10-bit window
    of steps i - 5 .. i + 4: fw = ctz(window >> 5 | 32), bk = ctz(bitreverse(window << 27) | 32).  Against the literal loops
    (reference src/scanRegistration.cpp:316-341: stop at the first long step, at most 5, inside the ring)."""
    rng = np.random.default_rng(11)
    for trial in range(300):
        n = int(rng.integers(1, 700))
        gap = rng.random(n) < rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])
        padded = ((n + 255) // 256) * 256                                  # the kernel ballots whole chunks of 256 steps
        bits = np.ones(64 + padded + 64, np.uint8)
        bits[64:64 + n - 1] = gap[:n - 1]                                  # steps 0 .. n - 2 exist
        words = np.packbits(bits, bitorder="little").view(np.uint32)
        for i in range(n):
            bit = i + 59
            lo, hi = int(words[bit >> 5]), int(words[(bit >> 5) + 1])
            win = (((hi << 32) | lo) >> (bit & 31)) & 0xffffffff           # v_alignbit_b32
            ctz = lambda x: (x & -x).bit_length() - 1
            brev = lambda x: int(format(x & 0xffffffff, "032b")[::-1], 2)
            fw = ctz((win >> 5) | 32)
            bk = ctz(brev((win << 27) & 0xffffffff) | 32)
            fw_ref = bk_ref = 0
            while fw_ref < 5 and i + fw_ref < n - 1 and not gap[i + fw_ref]: fw_ref += 1
            while bk_ref < 5 and i - 1 - bk_ref >= 0 and not gap[i - 1 - bk_ref]: bk_ref += 1
            assert (fw, bk) == (fw_ref, bk_ref), (trial, n, i)
