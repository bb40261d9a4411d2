"""Model checks (CPU, lane by lane in Python) of the decision logic the round-3 kernels rest on:

  * k_vox_lds (mapping_kernels.hip): pcl::VoxelGrid of one segment by one workgroup — RUN heads from the bounding-box independent
    cell test, one (voxel index, first point) key per run, a STABLE radix sort on the voxel index only (7-bit digits, wave-striped
    element ownership, per-(digit, wave) counts scanned digit-major, rank among the equal digits of a row), voxel heads, members
    added in input order — against the oracle's voxel filter (canonical order) bit for bit;
  * k_front (registration_kernels.hip): the decoupled look-back over the blocks of a sweep (per-ring counts, halfPassed index);
  * k_build_grids_fused (odometry_kernels.hip): three bucket tables as 16-bit counters packed two to a word — counting and the
    fetch-add of running offsets never carry from the low half into the high half while a cloud has at most 65535 points;
  * k_map_fit / k_map_solve (mapping_kernels.hip): valid factor records compacted per tile of 256 stack points and addressed densely
    through the tile prefixes.
They touch neither the GPU nor (except as the reference result of the first) the oracle."""
import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------ radix sort of the run keys
def radix_sort_model(vi, first, nt, capr, key_bits, rb=7):
    """radix_sort_pairs<NT, CAPR>: returns (vi, first) sorted; wave w owns elements [w * EPT * 64, (w + 1) * EPT * 64)."""
    nw, ept, nb = nt // 64, capr // nt, 1 << rb
    n = len(vi)
    hi = np.full(capr, 0xFFFFFFFF, np.uint64); lo = np.zeros(capr, np.uint64)
    hi[:n] = vi; lo[:n] = first
    for shift in range(0, key_bits, rb):
        rows_of = [min(ept, (n - w * ept * 64 + 63) >> 6) if n > w * ept * 64 else 0 for w in range(nw)]
        regs = {}                                                            # (wave, row, lane) -> (key, payload): the registers
        for w in range(nw):
            for k in range(rows_of[w]):
                for lane in range(64):
                    p = w * ept * 64 + k * 64 + lane
                    regs[(w, k, lane)] = (int(hi[p]), int(lo[p])) if p < n else (0xFFFFFFFF, 0)
        cnt = np.zeros((nb, nw), np.int64)                                   # digit-major, wave-minor
        for (w, k, lane), (key, _) in regs.items():
            cnt[(key >> shift) & (nb - 1), w] += 1
        base = (np.cumsum(cnt.reshape(-1)) - cnt.reshape(-1)).reshape(nb, nw)
        out_hi = np.full(capr, 0xFFFFFFFF, np.uint64); out_lo = np.zeros(capr, np.uint64)
        for w in range(nw):
            run = base[:, w].copy()
            for k in range(rows_of[w]):                                      # rows in order; inside a row the rank among equal digits = lower lanes first
                digits = [(regs[(w, k, lane)][0] >> shift) & (nb - 1) for lane in range(64)]
                seen = {}
                for lane in range(64):
                    d = digits[lane]
                    pos = run[d] + seen.get(d, 0)
                    seen[d] = seen.get(d, 0) + 1
                    out_hi[pos], out_lo[pos] = regs[(w, k, lane)]
                for d, c in seen.items():
                    run[d] += c
        hi, lo = out_hi, out_lo
    return hi[:n].astype(np.int64), lo[:n].astype(np.int64)


@pytest.mark.parametrize("nt,capr", [(64, 2048), (256, 8192), (1024, 24576)])
def test_radix_sort_of_run_keys_is_a_stable_sort_on_the_voxel_index(nt, capr):
    rng = np.random.default_rng(nt)
    for n in (1, 63, 64, 65, 700, capr // 3, capr - 1, capr):
        for bits in ((1, 7, 8, 14, 21, 31) if n <= 700 else (21,)):                  # (the big sizes once per geometry: three passes)
            vi = rng.integers(0, 1 << bits, n) if bits < 31 else rng.integers(0, (1 << 31) - 1, n)
            if n > 10:
                vi[rng.integers(0, n, n // 2)] = vi[rng.integers(0, n, n // 2)]      # plenty of repeated voxels
            first = np.arange(n) * 2 + 1                                             # ascending with the input order, like run starts
            s_vi, s_first = radix_sort_model(vi, first, nt, capr, bits)
            order = np.argsort(vi, kind="stable")
            assert np.array_equal(s_vi, vi[order]) and np.array_equal(s_first, first[order]), (nt, n, bits)


# ------------------------------------------------------------------------------------------------ k_vox_lds as a whole
def vox_lds_model(pts, leaf, nt):
    """The phases of k_vox_lds on one segment, f32 arithmetic as the kernel writes it; returns the centroids in output order."""
    f32 = np.float32
    p = pts.astype(f32)
    n = len(p)
    inv = f32(1.0) / f32(leaf)
    cell = np.floor(p[:, :3] * inv)                                                  # f32 products, floorf
    head = np.ones(n, bool)
    head[1:] = np.any(cell[1:] != cell[:-1], axis=1)                                 # pass 1: run heads, bounding-box independent
    gmn, gmx = p[:, :3].min(0), p[:, :3].max(0)
    d = ((gmx - gmn) * inv).astype(np.int64) + 1
    if int(d[0]) * int(d[1]) * int(d[2]) > 2147483647:
        return p.copy()
    minb = np.floor(gmn * inv).astype(np.int64)
    divb = np.floor(gmx * inv).astype(np.int64) - minb + 1
    fminb = minb.astype(f32)
    ijk = (cell - fminb).astype(np.int64)                                            # (int)(floorf(x * inv) - (float)min_b)
    vi_all = ijk[:, 0] + ijk[:, 1] * divb[0] + ijk[:, 2] * divb[0] * divb[1]
    firsts = np.nonzero(head)[0]
    cells = int(divb[0]) * int(divb[1]) * int(divb[2])
    key_bits = max(1, int(cells - 1).bit_length())
    capr = {64: 2048, 256: 8192, 1024: 24576}[nt]
    assert len(firsts) <= capr
    s_vi, s_first = radix_sort_model(vi_all[firsts], firsts, nt, capr, key_bits)
    out = []
    q = 0
    while q < len(s_vi):
        v = s_vi[q]
        acc = np.zeros(4, f32); cnt = 0
        while q < len(s_vi) and s_vi[q] == v:                                        # the runs of this voxel, ascending first point
            e = int(s_first[q])
            while True:
                acc = (acc + p[e]).astype(f32); cnt += 1                             # members added one by one, in input order
                e += 1
                if e >= n or head[e]:
                    break
            q += 1
        out.append(acc / f32(cnt))
    return np.array(out, f32)


@pytest.mark.parametrize("kind", ["ring", "cube", "random", "single"])
def test_lds_voxel_filter_model_equals_the_oracle_filter(O, kind):
    rng = np.random.default_rng(3)
    if kind == "ring":                                            # ring-ordered ground returns: long runs, voxels revisited by the next ring
        ang = np.concatenate([np.linspace(0, 2 * np.pi, 700, endpoint=False)] * 3)
        rad = np.repeat([9.0, 9.6, 10.3], 700) + rng.normal(0, 0.02, 2100)
        pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), -1.7 + rng.normal(0, 0.02, 2100), np.repeat([40.0, 41.0, 42.0], 700)], 1)
        leaf, nt = 0.8, 256
    elif kind == "cube":                                          # a map cube: one point per voxel from the last filter + unsorted new points
        pts = rng.uniform(-25, 25, (1500, 4)); pts[:, 2] *= 0.1
        pts = O.voxel_filter(pts.astype(np.float32), 0.4).astype(np.float64)
        pts = np.concatenate([pts, rng.uniform(-25, 25, (300, 4)) * [1, 1, 0.1, 1]])
        leaf, nt = 0.4, 64
    elif kind == "random":
        pts = rng.normal(0, 6, (5000, 4)); leaf, nt = 0.8, 256
    else:
        pts = np.array([[1.0, 2.0, 3.0, 4.0]]); leaf, nt = 0.4, 64
    got = vox_lds_model(pts, leaf, nt)
    ref = O.voxel_filter(pts.astype(np.float32), leaf, canonical=True)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), kind


# ------------------------------------------------------------------------------------------------ k_front's look-back over the blocks of a sweep
@pytest.mark.parametrize("R,nb,seed", [(16, 29, 0), (64, 128, 1), (128, 40, 2), (51, 7, 3), (64, 1, 4)])
def test_decoupled_lookback_over_the_blocks_of_a_sweep(R, nb, seed):
    """registration_kernels.hip front_lookback: every block publishes per ring one granule - first its own count (state A), then the inclusive prefix
    (state P) - and walks back over its predecessors, adding A's until it meets a P; slot R carries the halfPassed index as a running minimum.
    Blocks advance in a RANDOM interleaving (one granule read or one publication at a time, a block only ever started after its predecessors drew
    their tickets); whatever the schedule, every block ends up with the exclusive prefix over the blocks in front of it."""
    rng = np.random.default_rng(seed)
    hist = rng.integers(0, 40, (nb, R))
    half = np.where(rng.random(nb) < 0.2, rng.integers(0, 100000, nb), 0x7fffffff)
    want = np.cumsum(hist, axis=0) - hist
    want_half = np.concatenate([[0x7fffffff], np.minimum.accumulate(half)[:-1]])
    A, P = 1, 2
    state = np.zeros((nb, R + 1), np.int64)
    value = np.zeros((nb, R + 1), np.int64)
    got = np.zeros((nb, R + 1), np.int64)
    # per (block, slot) program counter: 0 = publish aggregate (or prefix for block 0), 1 = walking back at `t`, 2 = done
    pc = np.zeros((nb, R + 1), np.int64)
    t = np.zeros((nb, R + 1), np.int64)
    acc = np.zeros((nb, R + 1), np.int64)
    local = np.concatenate([hist, half[:, None]], axis=1)
    started = 0
    steps = 0
    while (pc < 2).any():
        steps += 1
        assert steps < 10_000_000
        if started < nb and (rng.random() < 0.3 or not ((pc[:started] < 2).any())):
            started += 1                                                          # the next ticket is drawn
            continue
        live = np.argwhere(pc[:started] < 2)
        blk, slot = live[rng.integers(len(live))]
        is_min = slot == R
        ident = 0x7fffffff if is_min else 0
        if pc[blk, slot] == 0:
            if blk == 0:
                state[blk, slot], value[blk, slot] = P, local[blk, slot]
                got[blk, slot] = ident
                pc[blk, slot] = 2
            else:
                state[blk, slot], value[blk, slot] = A, local[blk, slot]
                acc[blk, slot], t[blk, slot], pc[blk, slot] = ident, blk - 1, 1
        else:
            tt = t[blk, slot]
            if state[tt, slot] == 0:
                continue                                                          # not published yet: the lane spins
            v = value[tt, slot]
            acc[blk, slot] = min(acc[blk, slot], v) if is_min else acc[blk, slot] + v
            if state[tt, slot] == P or tt == 0:
                assert state[tt, slot] == P or tt > 0
                if state[tt, slot] != P:
                    t[blk, slot] = tt - 1
                    continue
                got[blk, slot] = acc[blk, slot]
                incl = min(acc[blk, slot], local[blk, slot]) if is_min else acc[blk, slot] + local[blk, slot]
                state[blk, slot], value[blk, slot] = P, incl
                pc[blk, slot] = 2
            else:
                t[blk, slot] = tt - 1
    assert np.array_equal(got[:, :R], want)
    assert np.array_equal(got[:, R], want_half)
    assert np.array_equal(value[nb - 1, :R], hist.sum(0))                          # what k_ring_starts reads: the ring lengths


# ------------------------------------------------------------------------------------------------ packed 16-bit bucket tables
def test_packed_16_bit_counters_never_carry_for_clouds_up_to_65535_points():
    rng = np.random.default_rng(9)
    H = 4096
    for n in (1, 1000, 65535):
        for skew in (False, True):
            h = rng.integers(0, H, n) if not skew else np.where(rng.random(n) < 0.9, 7, rng.integers(0, H, n))   # 90 % in one bucket
            words = np.zeros(H // 2, np.int64)
            for b in h:                                                           # count: atomicAdd(&tab[h >> 1], 1 << ((h & 1) * 16))
                words[b >> 1] = (words[b >> 1] + (1 << ((b & 1) * 16))) & 0xFFFFFFFF
            cnt = np.stack([words & 0xFFFF, words >> 16], 1).reshape(-1).astype(np.int64)
            assert np.array_equal(cnt, np.bincount(h, minlength=H))
            start = np.cumsum(cnt) - cnt                                          # scan: running offsets written back, two to a word
            words = (start[0::2] | (start[1::2] << 16)).astype(np.int64)
            pos = np.zeros(n, np.int64)
            for i, b in enumerate(h):                                             # fill: fetch-add returns the old word, the half is this point's position
                sh = (b & 1) * 16
                pos[i] = (int(words[b >> 1]) >> sh) & 0xFFFF
                words[b >> 1] = (int(words[b >> 1]) + (1 << sh)) & 0xFFFFFFFF
            assert len(np.unique(pos)) == n and pos.max() == n - 1
            order = np.argsort(pos)
            assert np.all(np.diff(h[order]) >= 0)                                 # bucket-contiguous, ascending bucket id
            assert np.array_equal(np.sort(pos), np.arange(n))


# ------------------------------------------------------------------------------------------------ tile-compacted factor records
def test_tile_compacted_records_are_addressed_densely_and_in_stack_order():
    rng = np.random.default_rng(4)
    for n in (0, 1, 255, 256, 257, 4380, 6352):
        valid = rng.random(n) < 0.3
        nt = (n + 255) >> 8
        store = np.full(max(1, nt) * 256, -1, np.int64)
        tile_cnt = np.zeros(nt, np.int64)
        for t in range(nt):                                                       # k_map_fit: rank of a valid point inside its tile of 256
            idx = np.arange(t * 256, min(n, t * 256 + 256))
            v = idx[valid[idx]]
            store[t * 256: t * 256 + len(v)] = v
            tile_cnt[t] = len(v)
        pre = np.concatenate([[0], np.cumsum(tile_cnt)])                          # k_map_solve: prefixes, then binary search per dense index
        got = []
        for d in range(int(pre[-1])):
            lo, hi = 0, nt - 1
            while lo < hi:
                mid = (lo + hi + 1) >> 1
                if pre[mid] <= d:
                    lo = mid
                else:
                    hi = mid - 1
            got.append(store[(lo << 8) + (d - pre[lo])])
        assert np.array_equal(np.array(got, np.int64), np.nonzero(valid)[0])
