"""Model checks of three lock-step building blocks of the HIP kernels, run lane by lane in Python:

  * wave_sweep (odometry_kernels.hip): 64 lanes own one bucket each; all candidates of all buckets are dealt out to the lanes,
    kSweep * 64 per round; the owner of a position is found by dropping lane ids into an LDS row at each bucket's first position
    and spreading them with an inclusive max-scan, with a carry between rows and rounds;
  * bitonic_sort_u64 (aloam_device.hpp): the all-ascending network with imaginary +inf slots beyond n and the stages inside
    aligned groups of eight keys fused;
  * the voxel centroids of k_ring_features: runs of consecutive same-voxel members, sorted by (voxel, first element), summed run
    by run in element order — against the plain definition (members of a voxel summed in input order, voxels ascending).
"""
import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------ wave_sweep
def sweep_model(cnt, rows_max):
    """cnt[64] bucket sizes -> list of (lane_that_loads, owner_bucket, offset_in_bucket) in the order the rounds deal them out."""
    incl = np.cumsum(cnt)
    total = int(incl[-1]); excl = incl - cnt
    dealt = []
    carry = 0
    base = 0
    while base < total:
        rows = rows_max if total - base > (rows_max - 1) * 64 else (total - base + 63) >> 6
        row = np.zeros(rows * 64, np.int64)
        for lane in range(64):
            slot = excl[lane] - base
            if cnt[lane] > 0 and 0 <= slot < rows * 64:
                row[slot] = lane + 1                                         # later lanes overwrite earlier ones only if they share a slot: impossible for cnt > 0
        for u in range(rows):
            own = row[u * 64:(u + 1) * 64].copy()
            if carry > own[0]: own[0] = carry
            own = np.maximum.accumulate(own)
            carry = int(own[63])
            for lane in range(64):
                i = base + u * 64 + lane
                if i < total:
                    o = int(own[lane]) - 1
                    dealt.append((lane, o, i - int(excl[o])))
        base += rows_max * 64
    return dealt


@pytest.mark.parametrize("rows_max", [2, 3])
def test_wave_sweep_deals_every_candidate_exactly_once(rows_max):
    rng = np.random.default_rng(5 + rows_max)
    for trial in range(400):
        style = trial % 5
        if style == 0:   cnt = rng.integers(0, 4, 64)
        elif style == 1: cnt = rng.integers(0, 2, 64) * rng.integers(0, 40, 64)
        elif style == 2: cnt = np.zeros(64, np.int64); cnt[rng.integers(0, 64)] = rng.integers(1, 700)
        elif style == 3: cnt = rng.integers(0, 30, 64)
        else:            cnt = np.zeros(64, np.int64)
        cnt = np.asarray(cnt, np.int64)
        if trial % 7 == 0: cnt[27:] = 0                                      # like the 27 look-ups of the fine block
        dealt = sweep_model(cnt, rows_max)
        want = [(b, k) for b in range(64) for k in range(int(cnt[b]))]
        assert [(o, k) for _, o, k in dealt] == want, (trial, cnt.tolist())


# ------------------------------------------------------------------------------------------------ bitonic
INF = float("inf")


def _cx(v, a, b):
    if v[a] > v[b]: v[a], v[b] = v[b], v[a]


def bitonic_model(x):
    n = len(x); npad = 1
    while npad < n: npad *= 2
    a = list(x) + [INF] * (max(npad, 8) - n)
    def groups(first):
        for g in range(0, max(npad, 8), 8):
            if g >= n: continue
            v = a[g:g + 8]
            if first:
                for b in (0, 2, 4, 6): _cx(v, b, b + 1)
                for b in (0, 4): _cx(v, b, b + 3); _cx(v, b + 1, b + 2)
                for b in (0, 2, 4, 6): _cx(v, b, b + 1)
                for o in range(4): _cx(v, o, 7 - o)
            else:
                for o in range(4): _cx(v, o, o + 4)
            for b in (0, 4): _cx(v, b, b + 2); _cx(v, b + 1, b + 3)
            for b in (0, 2, 4, 6): _cx(v, b, b + 1)
            a[g:g + 8] = v
    groups(True)
    k = 16
    while k <= npad:
        hk = k // 2
        for t in range(npad // 2):
            base = (t // hk) * k; off = t % hk
            i, l = base + off, base + (k - 1 - off)
            if l < n: _cx(a, i, l)
        j = k >> 2
        while j > 4:
            for t in range(npad // 2):
                i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); l = i + j
                if l < n: _cx(a, i, l)
            j >>= 1
        groups(False)
        k <<= 1
    return a[:n]


def test_bitonic_network_with_imaginary_padding_sorts_every_length():
    rng = np.random.default_rng(9)
    for n in list(range(0, 130)) + [255, 256, 257, 511, 700, 1023, 1024, 1025, 1477, 2047, 2048]:
        for rep in range(2):
            x = rng.integers(0, max(2, n // 2 if rep else n * 4), n).tolist() if n else []    # with and without duplicates
            assert bitonic_model(x) == sorted(x), n


# ------------------------------------------------------------------------------------------------ voxel runs
def test_voxel_centroids_from_sorted_runs_equal_the_plain_definition():
    rng = np.random.default_rng(13)
    for trial in range(200):
        L = int(rng.integers(1, 400))
        member = rng.random(L) < rng.choice([0.5, 0.9, 1.0])
        vi = np.cumsum(rng.integers(-1, 2, L) * (rng.random(L) < 0.3)) + 50                  # a walk that revisits voxels
        vi = np.where(rng.random(L) < 0.05, rng.integers(0, 100, L), vi)
        pts = rng.normal(size=(L, 4)).astype(np.float32) * 30
        # plain definition: voxels ascending, members of a voxel summed in input order (f32)
        want = []
        for v in sorted(set(vi[member].tolist())):
            s = np.zeros(4, np.float32); c = 0
            for e in range(L):
                if member[e] and vi[e] == v:
                    s = (s + pts[e]).astype(np.float32); c += 1
            want.append((s / np.float32(c)).astype(np.float32))
        # kernel scheme: run heads, runs sorted by (voxel, first element), continuation bits
        head = [member[e] and (e == 0 or not member[e - 1] or vi[e - 1] != vi[e]) for e in range(L)]
        cont = [member[e] and not head[e] for e in range(L)]
        runs = sorted((int(vi[e]), e) for e in range(L) if head[e])
        got = []
        p = 0
        while p < len(runs):
            v = runs[p][0]
            s = np.zeros(4, np.float32); c = 0
            q = p
            while q < len(runs) and runs[q][0] == v:
                e = runs[q][1]
                while True:
                    s = (s + pts[e]).astype(np.float32); c += 1
                    e += 1
                    if not (e < L and cont[e]): break
                q += 1
            got.append((s / np.float32(c)).astype(np.float32))
            p = q
        assert len(got) == len(want) and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, want)), trial


# ------------------------------------------------------------------------------------------------ DPP ladder reductions (round 2)
def dpp_row_shr(v, n, old):
    """v_mov_dpp row_shr:n with bound_ctrl off: lane i of every row of 16 takes lane i - n of the same row, `old` where there is none."""
    out = np.array(old, copy=True)
    for lane in range(64):
        src = lane - n
        if src // 16 == lane // 16 and src >= 0:
            out[lane] = v[src]
    return out


def dpp_row_bcast(v, which, row_mask, old):
    """row_bcast:15 (lane 15 of each row -> every lane of the next row) / row_bcast:31 (lane 31 -> rows 2 and 3), written only to the
    rows in row_mask; other lanes keep `old`."""
    out = np.array(old, copy=True)
    for lane in range(64):
        row = lane // 16
        if not (row_mask >> row) & 1:
            continue
        if which == 15 and row >= 1:
            out[lane] = v[16 * row - 1]
        elif which == 31 and row >= 2:
            out[lane] = v[31]
    return out


def wave_reduce_model(v, is_max):
    """wave_reduce_u32 (aloam_device.hpp): op(v, dpp(identity, v, ...)) for row_shr 1, 2, 4, 8, row_bcast15 (rows 1, 3), row_bcast31
    (rows 2, 3); the result is read from lane 63."""
    ident = 0 if is_max else 0xffffffff
    op = np.maximum if is_max else np.minimum
    v = np.asarray(v, np.uint64).copy()
    idv = np.full(64, ident, np.uint64)
    for n in (1, 2, 4, 8):
        v = op(v, dpp_row_shr(v, n, idv))
    v = op(v, dpp_row_bcast(v, 15, 0xA, idv))
    v = op(v, dpp_row_bcast(v, 31, 0xC, idv))
    return int(v[63])


def test_dpp_ladder_reduces_the_whole_wave_into_lane_63():
    rng = np.random.default_rng(17)
    for trial in range(500):
        v = rng.integers(0, 2 ** 32, 64, dtype=np.uint64)
        if trial % 5 == 0: v[:] = rng.integers(0, 3, 64)                     # many ties
        if trial % 7 == 0: v[rng.integers(0, 64)] = 0xffffffff
        assert wave_reduce_model(v, True) == int(v.max()) and wave_reduce_model(v, False) == int(v.min())
    for lane in range(64):                                                   # a single extreme in every possible lane
        v = np.full(64, 5, np.uint64); v[lane] = 9
        assert wave_reduce_model(v, True) == 9
        v[lane] = 1
        assert wave_reduce_model(v, False) == 1


def wave_min_packed_model(v):
    """wave_min_packed (aloam_device.hpp): reduce the high words; the low word by v_readlane when one lane holds the minimum, by a
    second ladder over the tied lanes otherwise."""
    hi = [int(x) >> 32 for x in v]; lo = [int(x) & 0xffffffff for x in v]
    mh = wave_reduce_model(hi, False)
    tied = [l for l in range(64) if hi[l] == mh]
    ml = lo[tied[0]] if len(tied) == 1 else wave_reduce_model([lo[l] if hi[l] == mh else 0xffffffff for l in range(64)], False)
    return (mh << 32) | ml


def test_packed_minimum_equals_the_64_bit_minimum():
    rng = np.random.default_rng(23)
    for trial in range(500):
        hi = rng.integers(0, 2 ** 32, 64, dtype=np.uint64)
        lo = rng.integers(0, 2 ** 32, 64, dtype=np.uint64)
        if trial % 3 == 0: hi[:] = rng.integers(0, 4, 64)                    # exactly equal distances: the tie-break decides
        v = [(int(h) << 32) | int(l) for h, l in zip(hi, lo)]
        if trial % 11 == 0: v = [0xffffffffffffffff] * 64                    # nobody holds a candidate
        if trial % 13 == 0: v[rng.integers(0, 64)] = 0xffffffffffffffff
        assert wave_min_packed_model(v) == min(v)
