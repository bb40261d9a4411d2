"""GPU parity tests: the HIP path, called through the C ABI (include/aloam_mi355x.h), against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for every integer / index / f32 feature array; poses within 1e-4 m / 1e-4 rad
(BASELINE.json north_star) — in practice they agree to ~1e-15 because the f64 sums only differ in association order.
"""
import glob
import os

import numpy as np
import pytest
import lm_scenarios
from conftest import bits_equal, quat_angle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4
FEATURES = ("cloud", "sharp", "less_sharp", "flat", "less_flat")


def _mk(binding, model, batch=1, max_points=140000, **kw):
    return binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field, batch=batch,
                         max_points=max_points, **kw)


def _assert_features_equal(fo, fg, ctx=""):
    for k in FEATURES:
        assert fo[k].shape == fg[k].shape, (ctx, k, fo[k].shape, fg[k].shape)
        assert bits_equal(fo[k], fg[k]), (ctx, k)


def _assert_pose_close(po, pg, ctx=""):
    assert np.abs(po["t_lc"] - pg["t_lc"]).max() < POSE_TOL_M and np.linalg.norm(po["t_w"] - pg["t_w"]) < POSE_TOL_M, (ctx, po, pg)
    assert quat_angle(po["q_lc"], pg["q_lc"]) < POSE_TOL_RAD and quat_angle(po["q_w"], pg["q_w"]) < POSE_TOL_RAD, (ctx, po, pg)


@pytest.mark.parametrize("name,frames,kw", [("VLP-16", 4, {}), ("HDL-32", 3, {}), ("HDL-64", 4, {}), ("HDL-64", 3, {"columns": 2200}),
                                            ("ROWS128", 4, {}), ("HDL-64", 4, {"rough": True}), ("VLP-16", 4, {"rough": True}),
                                            ("HDL-64", 3, {"rough": True, "nan_fraction": 0.02, "columns": 1500})])
def test_free_running_sequence_matches_oracle(O, binding, sequence, name, frames, kw):
    """Registration + odometry over consecutive sweeps, every intermediate array compared.  The `rough` cases are KITTI-shaped
    irregular sweeps: random no-returns, rings of unequal length, noisy "vegetation" sectors and verbatim repeated returns, which
    produce exactly equal curvatures — the HIP path orders ties by (curvature, index) like the oracle's canonical order (the
    reference's unstable std::sort leaves them implementation-defined, src/scanRegistration.cpp:288)."""
    scans, R, t, model = sequence(name, frames, seed=1, **kw)
    orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
    gpu = _mk(binding, model, max_points=max(len(s) for s in scans) + 64)
    for k, x in enumerate(scans):
        fo = orc.scan_register(x)
        gpu.scan_register(x)
        fg = gpu.features()
        _assert_features_equal(fo, fg, (name, k))
        so, co = orc.ring_ranges(); sg, cg = gpu.ring_ranges()
        assert np.array_equal(so, sg) and np.array_equal(co, cg)
        curv_o, lab_o, _ = orc.per_point(); curv_g, lab_g = gpu.per_point()
        sel = np.zeros(len(curv_o), bool)
        for s0, c0 in zip(so, co):
            if c0 >= 17:
                sel[s0 + 5:s0 + c0 - 6] = True                      # the indices the reference ever reads (scanRegistration.cpp:249-251)
        assert bits_equal(curv_o[sel], curv_g[sel]) and np.array_equal(lab_o[sel], lab_g[sel])
        po = orc.odometry_step()
        gpu.odometry_step()
        pg = gpu.pose()
        _assert_pose_close(po, pg, (name, k))
        so_, sg_ = orc.odom_stats(), gpu.odom_stats()
        for key in ("corner_corr", "plane_corr", "lm_iterations", "lm_successful", "termination"):
            assert so_[key] == sg_[key], (name, k, key, so_, sg_)
        assert np.allclose(so_["final_cost"], sg_["final_cost"], rtol=1e-9)
        if k > 0:
            eo, plo, eqo, pqo = orc.correspondences()
            eg, plg, eqg, pqg = gpu.correspondences()
            assert np.array_equal(eqo, eqg) and np.array_equal(pqo, pqg)
            assert bits_equal(eo.astype(np.float32), eg) and bits_equal(plo.astype(np.float32), plg)
        assert bits_equal(orc.cloud(O.CLOUD_CORNER_LAST), gpu.cloud(binding.CLOUD_CORNER_LAST))
        assert bits_equal(orc.cloud(O.CLOUD_SURF_LAST), gpu.cloud(binding.CLOUD_SURF_LAST))
    gpu.close()


@pytest.mark.parametrize("name,kw,cut", [("HDL-64", {"columns": 512}, 40), ("HDL-64", {}, 150)])
def test_sweeps_whose_first_ray_has_no_return(O, binding, sequence, name, kw, cut):
    """relTime is measured from the azimuth of the sweep's FIRST point (reference src/scanRegistration.cpp:141,211-214,239).  When the first rays of
    the first ring return nothing - any range-limited or real sweep - the points of the other rings that lie before that azimuth get a slightly
    NEGATIVE relTime, intensity = scanID - eps, and int(intensity) = scanID - 1: the reference's ring id of a few points in the middle of every
    ring's stretch is one too low, the last clouds are no longer ring-sorted, and its neighbour walks (src/laserOdometry.cpp:315-361,410-455)
    break where THOSE keys say.  The pair kernel handles such nearly-sorted clouds with the index-range form of the walk window
    (k_associate_nearly): same correspondences as the oracle's literal walks, bit for bit, and the sequence stays on the fast path."""
    scans, R, t, model = sequence(name, 4, seed=7, **kw)
    cut_scans = [x[cut:] for x in scans]                    # ring-major sweeps: the first `cut` rays of the first ring are gone
    orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range)
    gpu = _mk(binding, model, max_points=max(len(s) for s in scans) + 64)
    orders, want = [], []

    def order_of(cloud):
        key = cloud[:, 3].astype(np.int32)
        return 0 if not (np.diff(key) < 0).any() else (1 if (np.maximum.accumulate(key) - key).max() <= 2 else 2)

    searched = None
    for k, x in enumerate(cut_scans):
        fo = orc.scan_register(x)
        gpu.scan_register(x)
        _assert_features_equal(fo, gpu.features(), (name, k))
        if k > 0:
            want.append(searched)
        searched = (order_of(fo["less_sharp"]), order_of(fo["less_flat"]))     # what the NEXT step searches
        po = orc.odometry_step()
        gpu.odometry_step()
        _assert_pose_close(po, gpu.pose(), (name, k))
        so_, sg_ = orc.odom_stats(), gpu.odom_stats()
        for sk in ("corner_corr", "plane_corr", "lm_iterations", "lm_successful", "termination"):
            assert so_[sk] == sg_[sk], (name, k, sk, so_, sg_)
        if k > 0:
            eo, plo, eqo, pqo = orc.correspondences()
            eg, plg, eqg, pqg = gpu.correspondences()
            assert np.array_equal(eqo, eqg) and np.array_equal(pqo, pqg)
            assert bits_equal(eo.astype(np.float32), eg) and bits_equal(plo.astype(np.float32), plg)
            orders.append(gpu.last_cloud_order())
    assert orders == want and sum(1 for o in want if o == (1, 1)) >= 2, (orders, want)   # nearly ring-sorted clouds stay with the pair kernel (order 1), not the literal walks (2)
    gpu.close()


def test_batched_sequences_are_independent_and_match_oracle(O, binding, sequence):
    """batch = 3 different sequences in one context == three single oracle runs (no cross-talk between sequences)."""
    seqs = [sequence("HDL-64", 3, seed=s, columns=1024) for s in (11, 12, 13)]
    model = seqs[0][3]
    gpu = _mk(binding, model, batch=3, max_points=64 * 1024)
    orcs = [O.Oracle(n_scans=64, min_range=model.min_range) for _ in seqs]
    for k in range(3):
        gpu.scan_register([s[0][k] for s in seqs])
        for b, orc in enumerate(orcs):
            fo = orc.scan_register(seqs[b][0][k])
            _assert_features_equal(fo, gpu.features(b), (b, k))
        gpu.odometry_step()
        for b, orc in enumerate(orcs):
            _assert_pose_close(orc.odometry_step(), gpu.pose(b), (b, k))
    gpu.close()


def test_teacher_forced_odometry_step(O, binding, sequence):
    """Odometry alone: features, last clouds and warm start injected from the oracle (isolates stage 2)."""
    scans, R, t, model = sequence("HDL-64", 3, seed=4, columns=1024)
    orc = O.Oracle(n_scans=64, min_range=model.min_range)
    feats = []
    for x in scans:
        feats.append(orc.scan_register(x))
        orc.odometry_step()
    para_q, para_t = np.array([0.001, -0.002, 0.012, 0.9999]), np.array([0.9, 0.05, -0.01])
    para_q /= np.linalg.norm(para_q)
    q_w, t_w = np.array([0.0, 0.0, 0.1, 0.995]), np.array([3.0, 1.0, 0.2])
    o2 = O.Oracle(n_scans=64, min_range=model.min_range)
    gpu = _mk(binding, model, max_points=70000)
    for dev in (o2, gpu):
        dev.set_features(feats[2])
        dev.set_last(feats[1]["less_sharp"], feats[1]["less_flat"])
        dev.set_state(para_q, para_t, q_w, t_w, inited=True)
        dev.odometry_step()
    _assert_pose_close(o2.pose(), gpu.pose())
    assert o2.odom_stats()["corner_corr"] == gpu.odom_stats()["corner_corr"]
    assert np.abs(gpu.pose()["t_lc"] - para_t).max() > 1e-3               # the solve actually moved the estimate
    gpu.close()


def test_sweeps_in_firing_order_through_the_one_pass_front_end(O, binding, sequence):
    """The message order of a real Velodyne driver: all lasers of a firing, then the next azimuth - every wave of k_front sees 64 different rings,
    every 1024-point block adds 16 points to each ring's slab, and halfPassed (src/scanRegistration.cpp:220-223) flips in the MIDDLE of the sweep,
    i.e. in a block whose predecessors carry it through the look-back.  Three different sweeps in one batch (the blocks of several sweeps interleave
    in the launch), each against the oracle: clouds, ring ranges and features bit for bit, and the dense full cloud only made on request."""
    scans, _, _, model = sequence("HDL-64", 3, seed=5, columns=512)
    R = model.n_scans
    fired = []
    for x in scans:
        assert len(x) == R * 512                                           # the arena returns every ray: ring-major [R][512]
        fired.append(np.ascontiguousarray(x.reshape(R, 512, 4).transpose(1, 0, 2).reshape(-1, 4)))
    fired[2] = fired[2][: 23 * 1024 + 517]                                 # a ragged last block; rings of unequal length
    gpu = _mk(binding, model, batch=3, max_points=R * 512 + 64)
    gpu.scan_register(fired)
    for b, x in enumerate(fired):
        orc = O.Oracle(n_scans=R, min_range=model.min_range)
        fo = orc.scan_register(x)
        _assert_features_equal(fo, gpu.features(b), ("fired", b))
        so, co = orc.ring_ranges(); sg, cg = gpu.ring_ranges(b)
        assert np.array_equal(so, sg) and np.array_equal(co, cg)
    gpu.close()


@pytest.mark.parametrize("name,columns,seed", [("VLP-16", 300, 11), ("HDL-32", 260, 12), ("HDL-64", 200, 13), ("HDL-64", 390, 14)])
def test_front_end_on_shuffled_truncated_and_holed_sweeps(O, binding, sequence, name, columns, seed):
    """k_front against the reference's sequential loop on inputs no sensor would send: the same sweep in ring-major order, in a RANDOM order (every
    wave sees up to 64 rings, the stable rank of a point depends on every block in front of it, halfPassed flips wherever the order says), cut
    off in the middle of a block, and with NaN / too-close returns sprinkled in - four different sequences in one batch, so that the tickets and
    look-back granules of several sweeps interleave.  Clouds, ring ranges and the four feature clouds bit for bit."""
    scans, _, _, model = sequence(name, 1, seed=seed, columns=columns)
    x = scans[0]
    rng = np.random.default_rng(seed)
    variants = [x, x[rng.permutation(len(x))], x[: len(x) // 2 + 333], x.copy()]
    holes = rng.random(len(x)) < 0.05
    variants[3][holes, 0] = np.nan
    variants[3][rng.random(len(x)) < 0.03, :3] *= 1e-3                      # inside minimum_range
    variants[3] = variants[3][rng.permutation(len(x))]
    gpu = _mk(binding, model, batch=4, max_points=len(x) + 64)
    for rep in range(2):                                                   # twice: the second launch reuses tickets and granules (epoch tags)
        gpu.scan_register(variants if rep == 0 else variants[::-1])
        for b, v in enumerate(variants if rep == 0 else variants[::-1]):
            orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
            fo = orc.scan_register(v)
            _assert_features_equal(fo, gpu.features(b), (name, rep, b))
            so, co = orc.ring_ranges(); sg, cg = gpu.ring_ranges(b)
            assert np.array_equal(so, sg) and np.array_equal(co, cg), (name, rep, b)
    gpu.close()


def test_dense_full_cloud_is_made_on_demand_and_mapping_reads_the_slabs(O, binding, sequence):
    """The device keeps the registered sweep as one slab per ring; the reference's dense laserCloud (src/scanRegistration.cpp:246-252) is assembled when
    somebody asks for it.  (i) Replacing ONE sequence's full cloud from outside must leave the other sequence's cloud of the last registration in
    place; (ii) /velodyne_cloud_registered (src/laserMapping.cpp:836-846) is the same whether laserMapping's full-resolution input came straight
    from the slabs (nobody asked for the dense cloud before the step) or from the dense copy (somebody did)."""
    scans, _, _, model = sequence("HDL-64", 2, seed=6, columns=512)
    orc = [O.Oracle(n_scans=model.n_scans, min_range=model.min_range) for _ in range(2)]
    want = [orc[b].scan_register(scans[b])["cloud"] for b in range(2)]
    gpu = _mk(binding, model, batch=2, max_points=len(scans[0]) + 64)
    gpu.scan_register([scans[0], scans[1]])
    other = np.ascontiguousarray(want[1][:1000][::-1])
    gpu.set_full_cloud(other, seq=0)                                       # before anybody read the dense cloud
    assert bits_equal(gpu.cloud(binding.CLOUD_FULL, 0), other)
    assert bits_equal(gpu.cloud(binding.CLOUD_FULL, 1), want[1])
    gpu.close()
    reg = []
    for ask_first in (False, True):
        g = _mk(binding, model, batch=2, max_points=len(scans[0]) + 64)
        g.mapping_enable(0.4, 0.8, pool_points=65536)
        for k in range(2):
            g.scan_register([scans[k], scans[1 - k]])
            g.odometry_step()
            if ask_first:
                assert bits_equal(g.cloud(binding.CLOUD_FULL, 0), want[k])
            g.mapping_step()
        reg.append([g.map_cloud(binding.MAP_REGISTERED, b) for b in range(2)])
        g.close()
    for b in range(2):
        assert len(reg[0][b]) == len(want[1 - b]) and bits_equal(reg[0][b], reg[1][b]), b


def test_input_layouts_and_nan_filter(O, binding, syn, sequence):
    """stride-32 PointCloud2 records == stride-16; NaN returns are dropped like removeNaNFromPointCloud does."""
    scans, R, t, model = sequence("VLP-16", 1, seed=8, nan_fraction=0.03)
    x = scans[0]
    assert np.isnan(x[:, 0]).sum() > 100
    orc = O.Oracle(n_scans=16, min_range=model.min_range)
    fo = orc.scan_register(x)
    gpu = _mk(binding, model, max_points=40000)
    gpu.scan_register(x)
    _assert_features_equal(fo, gpu.features(), "nan")
    wide = np.zeros((len(x), 8), np.float32); wide[:, :4] = x; wide[:, 4:] = 123.0
    gpu.scan_register(wide)
    _assert_features_equal(fo, gpu.features(), "stride32")
    # 12-byte records (x, y, z): the reference never reads the 4th float of its input (src/scanRegistration.cpp:132-133,239)
    gpu.scan_register(np.ascontiguousarray(x[:, :3]))
    _assert_features_equal(fo, gpu.features(), "stride12")
    gpu.close()
    g2 = binding.Aloam(n_scans=16, min_range=model.min_range, ring_from_field=True, max_points=40000)
    with pytest.raises(binding.AloamError) as e:                           # ring_from_field IS the 4th float
        g2.scan_register(np.ascontiguousarray(x[:, :3]))
    assert e.value.code == binding.E_ARG
    g2.close()


def test_edge_cases(O, binding):
    gpu = binding.Aloam(n_scans=16, min_range=0.3, max_points=4096)
    for bad in (np.zeros((0, 4), np.float32), np.full((10, 4), np.nan, np.float32), np.full((10, 4), 0.01, np.float32)):
        with pytest.raises(binding.AloamError) as e:
            gpu.scan_register(bad)
        assert e.value.code == binding.E_EMPTY
    with pytest.raises(binding.AloamError) as e:                          # larger than max_points
        gpu.scan_register(np.ones((5000, 4), np.float32))
    assert e.value.code == binding.E_CAPACITY
    # rings too short to select anything: cloud is produced, no features, odometry still steps
    tiny = np.array([[5, 0, 0, 0], [5, 1, 0, 0], [5, 2, 0.2, 0]], np.float32)
    orc = O.Oracle(16, 0.3)
    fo = orc.scan_register(tiny)
    gpu.scan_register(tiny)
    _assert_features_equal(fo, gpu.features(), "tiny")
    gpu.odometry_step(); gpu.scan_register(tiny); gpu.odometry_step()
    assert gpu.odom_stats()["termination"] == [4, 4]                      # no residuals
    assert np.array_equal(gpu.pose()["q_w"], [0, 0, 0, 1])
    gpu.close()
    # a ring longer than max_ring_points is reported, not silently truncated
    g2 = binding.Aloam(n_scans=16, min_range=0.3, max_points=8192, max_ring_points=2059)
    ang = np.linspace(0, -2 * np.pi, 3000, endpoint=False)
    ring = np.stack([10 * np.cos(ang), 10 * np.sin(ang), np.zeros_like(ang), np.zeros_like(ang)], 1).astype(np.float32)
    with pytest.raises(binding.AloamError) as e:
        g2.scan_register(ring)
    assert e.value.code == binding.E_CAPACITY
    g2.close()
    # odometry before any features
    g3 = binding.Aloam(n_scans=16, min_range=0.3, max_points=4096)
    with pytest.raises(binding.AloamError) as e:
        g3.odometry_step()
    assert e.value.code == binding.E_STATE
    g3.close()


def test_single_long_ring_uses_large_lds_class(O, binding):
    """One ring of 3000 points (> 2059): handled by the 4096-key LDS class, bit-exact."""
    rng = np.random.default_rng(0)
    ang = np.linspace(np.pi, -np.pi, 3000, endpoint=False)
    rad = 10 + 3 * np.sign(np.sin(6 * ang)) + rng.normal(size=3000) * 0.02
    ring = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.tan(np.deg2rad(1.0)) * rad, np.zeros_like(ang)], 1).astype(np.float32)
    orc = O.Oracle(16, 0.3)
    fo = orc.scan_register(ring)
    gpu = binding.Aloam(n_scans=16, min_range=0.3, max_points=4096, max_ring_points=4107)
    gpu.scan_register(ring)
    _assert_features_equal(fo, gpu.features(), "long ring")
    assert len(fo["sharp"]) > 0 and len(fo["less_flat"]) > 100
    gpu.close()


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("ref")))
def test_gpu_reproduces_committed_goldens(binding, path):
    g = np.load(path)
    gpu = binding.Aloam(n_scans=int(g["n_scans"]), min_range=float(g["min_range"]), max_points=40000)
    k = 0
    while f"scan{k}" in g:
        gpu.scan_register(g[f"scan{k}"])
        f = gpu.features()
        for key in ("sharp", "less_sharp", "flat", "less_flat"):
            assert bits_equal(f[key], g[f"{key}{k}"]), (path, k, key)
        assert bits_equal(f["cloud"][:, 3], g[f"cloud_intensity{k}"])
        gpu.odometry_step()
        p = gpu.pose()
        assert np.abs(p["t_lc"] - g[f"t_lc{k}"]).max() < POSE_TOL_M and quat_angle(p["q_lc"], g[f"q_lc{k}"]) < POSE_TOL_RAD
        assert np.linalg.norm(p["t_w"] - g[f"t_w{k}"]) < POSE_TOL_M and quat_angle(p["q_w"], g[f"q_w{k}"]) < POSE_TOL_RAD
        k += 1
    gpu.close()


def test_full_size_batch_properties(O, binding, syn):
    """BASELINE size (HDL-64, 131072 points per sweep) in a batch of 8 through the device-resident entry point:
    size-independent properties + every sequence checked against its own oracle run."""
    import torch
    B, T = 8, 3
    dev = torch.device("cuda", 0)
    model = syn.sensor_model("HDL-64", device=dev)
    NP = model.dirs.shape[0]
    data = torch.zeros((B, T, NP, 4), dtype=torch.float32, device=dev)
    counts = np.zeros((B, T), np.int32)
    world = syn.make_world(321).to(dev)
    for b in range(B):
        R, t = syn.trajectory(T, seed=40 + b, start_angle=0.5 * b)
        gen = torch.Generator(device=dev).manual_seed(40 + b)
        for k in range(T):
            s = syn.render_scan(world, model, R[k], t[k], 0.02, gen)
            counts[b, k] = len(s); data[b, k, :len(s)] = s
    torch.cuda.synchronize()
    gpu = _mk(binding, model, batch=B, max_points=NP, max_ring_points=2059)
    host = data.cpu().numpy()
    orcs = [O.Oracle(n_scans=64, min_range=model.min_range) for _ in range(B)]
    for k in range(T):
        gpu.process_device(data.data_ptr() + k * NP * 16, T * NP * 16, counts[:, k])
        gpu.synchronize()
        for b in range(B):
            orcs[b].scan_register(host[b, k, :counts[b, k]])
            po = orcs[b].odometry_step()
            _assert_pose_close(po, gpu.pose(b), (b, k))
            so_, sg_ = orcs[b].odom_stats(), gpu.odom_stats(b)
            for key in ("corner_corr", "plane_corr", "lm_iterations", "lm_successful", "termination"):
                assert so_[key] == sg_[key], (b, k, key, so_, sg_)
            assert bits_equal(orcs[b].cloud(O.CLOUD_SURF_LAST), gpu.cloud(binding.CLOUD_SURF_LAST, b)), (b, k)
            assert bits_equal(orcs[b].cloud(O.CLOUD_CORNER_LAST), gpu.cloud(binding.CLOUD_CORNER_LAST, b)), (b, k)
        for b in range(B):
            cl = gpu.cloud(binding.CLOUD_FULL, b)
            s0, c0 = gpu.ring_ranges(b)
            assert c0.sum() == len(cl) and c0[51:].sum() == 0 and len(cl) > 90000
            assert np.all(np.floor(cl[:, 3] + 0.06) == np.repeat(np.arange(64), c0))          # ring-sorted, stable compaction
            sharp, less = gpu.cloud(binding.CLOUD_SHARP, b), gpu.cloud(binding.CLOUD_CORNER_LAST, b)
            assert {p.tobytes() for p in sharp} <= {p.tobytes() for p in less}
            assert len(sharp) <= 12 * 51 and len(gpu.cloud(binding.CLOUD_FLAT, b)) <= 24 * 51
    # the estimated motion is ~1 m per sweep for every sequence
    for b in range(B):
        assert 0.8 < np.linalg.norm(gpu.pose(b)["t_lc"]) < 1.2
    # idempotence: replaying the last sweep into a fresh context reproduces its registration bit-for-bit
    g2 = _mk(binding, model, batch=B, max_points=NP, max_ring_points=2059)
    g2.scan_register_device(data.data_ptr() + (T - 1) * NP * 16, T * NP * 16, counts[:, T - 1])
    g2.synchronize()
    for b in range(B):
        assert bits_equal(g2.cloud(binding.CLOUD_LESS_FLAT, b), gpu.cloud(binding.CLOUD_SURF_LAST, b))
    g2.close(); gpu.close()


def test_distortion_mode_matches_oracle(O, binding, sequence):
    """DISTORTION 1 (compiled out in the reference's nodes, src/laserOdometry.cpp:59): per-point interpolation ratio in
    TransformToStart and in the factors.  Same bar as the shipped mode: correspondences identical (indices and coordinates bit
    for bit), solver statistics identical, poses to 1e-9."""
    for name, frames, kw in (("HDL-64", 4, {"columns": 1024}), ("VLP-16", 4, {}), ("HDL-64", 3, {"columns": 1024, "rough": True})):
        scans, R, t, model = sequence(name, frames, seed=21, **kw)
        orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range, distortion=True)
        ref = O.Oracle(n_scans=model.n_scans, min_range=model.min_range)      # s = 1, to show the mode does something
        gpu = _mk(binding, model, max_points=70000, distortion=True)
        moved = 0.0
        for k, x in enumerate(scans):
            _assert_features_equal(orc.scan_register(x), (gpu.scan_register(x), gpu.features())[1], ("distortion", name, k))
            ref.scan_register(x)
            po, pr = orc.odometry_step(), ref.odometry_step()
            gpu.odometry_step()
            pg = gpu.pose()
            assert np.abs(po["t_lc"] - pg["t_lc"]).max() < 1e-9 and quat_angle(po["q_lc"], pg["q_lc"]) < 1e-9, (name, k, po, pg)
            assert np.abs(po["t_w"] - pg["t_w"]).max() < 1e-9 and quat_angle(po["q_w"], pg["q_w"]) < 1e-9, (name, k, po, pg)
            so_, sg_ = orc.odom_stats(), gpu.odom_stats()
            for key in ("corner_corr", "plane_corr", "lm_iterations", "lm_successful", "termination"):
                assert so_[key] == sg_[key], (name, k, key, so_, sg_)
            if k > 0:
                eo, plo, eqo, pqo = orc.correspondences()
                eg, plg, eqg, pqg = gpu.correspondences()
                assert np.array_equal(eqo, eqg) and np.array_equal(pqo, pqg), (name, k)
                assert bits_equal(eo.astype(np.float32), eg) and bits_equal(plo.astype(np.float32), plg), (name, k)
            moved = max(moved, np.abs(po["t_lc"] - pr["t_lc"]).max())
        assert moved > 1e-4                                                   # not the s = 1 solution
        gpu.close()


@pytest.mark.parametrize("mode", ["unsorted", "far"])
def test_last_clouds_that_are_not_ring_sorted_or_out_of_range(O, binding, sequence, mode):
    """aloam_set_last accepts any cloud.  'unsorted': ring keys not ascending -> the literal walk loops (:312-361 / :402-455)
    instead of the ring-key window; 'far': a coordinate beyond the range the cell arithmetic is exact for -> literal brute-force
    1-NN as well.  Both must give the oracle's correspondences (the oracle always walks literally)."""
    scans, R, t, model = sequence("HDL-64", 3, seed=8, columns=1024)
    orc = O.Oracle(n_scans=64, min_range=model.min_range)
    feats = []
    for x in scans:
        feats.append(orc.scan_register(x))
        orc.odometry_step()
    rng = np.random.default_rng(3)
    corner, surf = feats[1]["less_sharp"].copy(), feats[1]["less_flat"].copy()
    if mode == "unsorted":
        for c in (corner, surf):                                              # swap two blocks of rings: keys go down once
            k = len(c) // 2
            c[:] = np.concatenate([c[k:], c[:k]])
    else:
        surf[rng.integers(len(surf))][:3] = (5000.0, 10.0, 1.0)
        corner[rng.integers(len(corner))][:3] = (-4200.0, 0.0, 0.0)
    para_q, para_t = np.array([0.0, 0.0, 0.01, 1.0]), np.array([0.95, 0.02, 0.0])
    para_q /= np.linalg.norm(para_q)
    o2 = O.Oracle(n_scans=64, min_range=model.min_range)
    gpu = _mk(binding, model, max_points=70000)
    for dev in (o2, gpu):
        dev.set_features(feats[2])
        dev.set_last(corner, surf)
        dev.set_state(para_q, para_t, [0, 0, 0, 1.0], [0, 0, 0.0], inited=True)
        dev.odometry_step()
    eo, plo, eqo, pqo = o2.correspondences()
    eg, plg, eqg, pqg = gpu.correspondences()
    assert np.array_equal(eqo, eqg) and np.array_equal(pqo, pqg)
    assert bits_equal(eo.astype(np.float32), eg) and bits_equal(plo.astype(np.float32), plg)
    _assert_pose_close(o2.pose(), gpu.pose(), mode)
    assert o2.odom_stats()["plane_corr"] == gpu.odom_stats()["plane_corr"] and min(o2.odom_stats()["plane_corr"]) > 100
    gpu.close()


def test_flagged_sequences_among_normal_ones_in_one_batch(O, binding, sequence):
    """k_associate_pair serves the sequences whose last clouds are ring-sorted and in range, k_associate_flagged (one workgroup per
    sequence, returns at once otherwise) the others — in ONE launch pair per class.  A batch of five with two flagged sequences (one not
    ring-sorted, one with far coordinates) between three normal ones: every sequence must give its own oracle's correspondences and pose."""
    scans, R, t, model = sequence("HDL-64", 3, seed=8, columns=1024)
    orc = O.Oracle(n_scans=64, min_range=model.min_range)
    feats = []
    for x in scans:
        feats.append(orc.scan_register(x))
        orc.odometry_step()
    rng = np.random.default_rng(4)
    base_c, base_s = feats[1]["less_sharp"], feats[1]["less_flat"]
    lasts = []
    for mode in ("normal", "unsorted", "normal", "far", "normal"):
        c, sf = base_c.copy(), base_s.copy()
        if mode == "unsorted":
            c, sf = np.concatenate([c[len(c) // 2:], c[:len(c) // 2]]), np.concatenate([sf[len(sf) // 2:], sf[:len(sf) // 2]])
        elif mode == "far":
            sf[rng.integers(len(sf))][:3] = (5000.0, 10.0, 1.0)
            c[rng.integers(len(c))][:3] = (-4200.0, 0.0, 0.0)
        lasts.append((c, sf))
    starts = [(np.array([0.0, 0.0, 0.01 * (b - 2), 1.0]), np.array([0.95 + 0.01 * b, 0.02, 0.0])) for b in range(5)]
    gpu = _mk(binding, model, batch=5, max_points=70000)
    oracles = []
    for b, ((c, sf), (pq, pt)) in enumerate(zip(lasts, starts)):
        pq = pq / np.linalg.norm(pq)
        o2 = O.Oracle(n_scans=64, min_range=model.min_range)
        o2.set_features(feats[2]); o2.set_last(c, sf); o2.set_state(pq, pt, [0, 0, 0, 1.0], [0, 0, 0.0], inited=True)
        o2.odometry_step()
        oracles.append(o2)
        gpu.set_features(feats[2], seq=b); gpu.set_last(c, sf, seq=b); gpu.set_state(pq, pt, [0, 0, 0, 1.0], [0, 0, 0.0], seq=b, inited=True)
    gpu.odometry_step()
    for b, o2 in enumerate(oracles):
        eo, plo, eqo, pqo = o2.correspondences()
        eg, plg, eqg, pqg = gpu.correspondences(b)
        assert np.array_equal(eqo, eqg) and np.array_equal(pqo, pqg), b
        assert bits_equal(eo.astype(np.float32), eg) and bits_equal(plo.astype(np.float32), plg), b
        _assert_pose_close(o2.pose(), gpu.pose(b), b)
        assert min(o2.odom_stats()["plane_corr"]) > 100
    gpu.close()


@pytest.mark.parametrize("outer,lm", [(1, 4), (3, 2), (2, 8), (2, 0)])
def test_solver_settings_other_than_the_reference_defaults(O, binding, sequence, outer, lm):
    """opti_counter loop (src/laserOdometry.cpp:278) and options.max_num_iterations (:496) are configuration here; the
    device loop must follow the oracle for other values too (lm = 0: evaluation only, the pose stays at the warm start)."""
    scans, R, t, model = sequence("HDL-64", 3, seed=9, columns=512)
    orc = O.Oracle(n_scans=64, min_range=model.min_range, outer_iterations=outer, lm_max_iterations=lm)
    gpu = _mk(binding, model, max_points=40000, outer_iterations=outer, lm_max_iterations=lm)
    for k, x in enumerate(scans):
        _assert_features_equal(orc.scan_register(x), (gpu.scan_register(x), gpu.features())[1], (outer, lm, k))
        po = orc.odometry_step()
        gpu.odometry_step()
        _assert_pose_close(po, gpu.pose(), (outer, lm, k))
        so_, sg_ = orc.odom_stats(), gpu.odom_stats()
        for key in ("corner_corr", "plane_corr", "lm_iterations", "lm_successful", "termination"):
            assert so_[key] == sg_[key], (outer, lm, k, key, so_, sg_)
    gpu.close()


def test_host_batch_entry_is_double_buffered_and_matches_device_entry(O, binding, syn):
    """aloam_process_host (one pinned host buffer for the whole batch, batched H2D copy on the copy stream, two device slabs) over
    several back-to-back asynchronous steps == aloam_process_device on the same sweeps, bit for bit, and == the oracle."""
    import torch
    B, T = 3, 5
    dev = torch.device("cuda", 0)
    model = syn.sensor_model("HDL-64", columns=512, device=dev)
    NP = model.dirs.shape[0]
    data = torch.zeros((B, T, NP, 4), dtype=torch.float32, device=dev)
    counts = np.zeros((B, T), np.int32)
    world = syn.make_world(77).to(dev)
    for b in range(B):
        R, t = syn.trajectory(T, seed=90 + b, start_angle=0.3 * b)
        gen = torch.Generator(device=dev).manual_seed(90 + b)
        for k in range(T):
            s = syn.render_scan(world, model, R[k], t[k], 0.02, gen)
            counts[b, k] = len(s); data[b, k, :len(s)] = s
    host = data.cpu().pin_memory()
    g_dev = _mk(binding, model, batch=B, max_points=NP, max_ring_points=2059)
    g_host = _mk(binding, model, batch=B, max_points=NP, max_ring_points=2059)
    for k in range(T):                                                     # no synchronisation in between: the slabs alternate
        g_host.process_host(host.data_ptr() + k * NP * 16, T * NP * 16, counts[:, k])
        g_dev.process_device(data.data_ptr() + k * NP * 16, T * NP * 16, counts[:, k])
    g_host.input_consumed()
    g_host.synchronize(); g_dev.synchronize()
    orcs = [O.Oracle(n_scans=64, min_range=model.min_range) for _ in range(B)]
    hn = host.numpy()
    for b in range(B):
        for k in range(T):
            orcs[b].scan_register(hn[b, k, :counts[b, k]])
            po = orcs[b].odometry_step()
        ph, pd = g_host.pose(b), g_dev.pose(b)
        for key in ("q_w", "t_w", "q_lc", "t_lc"):
            assert np.array_equal(ph[key], pd[key]), (b, key)
        _assert_pose_close(po, ph, b)
        assert bits_equal(g_host.cloud(binding.CLOUD_SURF_LAST, b), g_dev.cloud(binding.CLOUD_SURF_LAST, b))
        assert bits_equal(orcs[b].cloud(O.CLOUD_SURF_LAST), g_host.cloud(binding.CLOUD_SURF_LAST, b))
    # the 12-byte wire format through the same entry: same poses bit for bit
    host12 = data[..., :3].contiguous().cpu().pin_memory()
    g12 = _mk(binding, model, batch=B, max_points=NP, max_ring_points=2059)
    for k in range(T):
        g12.process_host(host12.data_ptr() + k * NP * 12, T * NP * 12, counts[:, k], 12)
    g12.synchronize()
    for b in range(B):
        for key in ("q_w", "t_w", "q_lc", "t_lc"):
            assert np.array_equal(g12.pose(b)[key], g_dev.pose(b)[key]), (b, key)
        assert bits_equal(g12.cloud(binding.CLOUD_SURF_LAST, b), g_dev.cloud(binding.CLOUD_SURF_LAST, b))
    g12.close()
    # a pageable single-sequence buffer through the same entry (the runtime stages it synchronously)
    g1 = _mk(binding, model, batch=1, max_points=NP, max_ring_points=2059)
    x = np.ascontiguousarray(hn[0, 0, :counts[0, 0]])
    g1.scan_register_host(x.ctypes.data, 0, [len(x)])
    g1.synchronize()
    o1 = O.Oracle(n_scans=64, min_range=model.min_range)
    _assert_features_equal(o1.scan_register(x), g1.features(), "pageable host")
    g1.close(); g_host.close(); g_dev.close()


def test_stage_sized_contexts(O, binding, sequence):
    """aloam_create_stages: a context that hosts one stage (what each ROS node shim creates) gives the same results as the full
    one through the hand-over the topics allow, and refuses the other stages' entry points with ALOAM_E_STATE."""
    scans, R, t, model = sequence("VLP-16", 3, seed=6)
    full = _mk(binding, model, max_points=40000)
    reg = _mk(binding, model, max_points=40000, stages=binding.STAGE_REGISTRATION)
    odo = _mk(binding, model, max_points=40000, stages=binding.STAGE_ODOMETRY)
    mp = _mk(binding, model, max_points=40000, stages=binding.STAGE_MAPPING)
    fullm = _mk(binding, model, max_points=40000)
    fullm.mapping_enable(0.2, 0.4, 65536); mp.mapping_enable(0.2, 0.4, 65536)
    for x in scans:
        full.scan_register(x); reg.scan_register(x)
        f = reg.features()
        _assert_features_equal(full.features(), f, "registration-only context")
        odo.set_features(f)
        full.odometry_step(); odo.odometry_step()
        pf, po = full.pose(), odo.pose()
        for key in ("q_w", "t_w", "q_lc", "t_lc"):
            assert np.array_equal(pf[key], po[key]), key
        corner, surf = odo.cloud(binding.CLOUD_CORNER_LAST), odo.cloud(binding.CLOUD_SURF_LAST)
        assert bits_equal(corner, full.cloud(binding.CLOUD_CORNER_LAST)) and bits_equal(surf, full.cloud(binding.CLOUD_SURF_LAST))
        a = mp.mapping_step_inputs(po["q_w"], po["t_w"], corner, surf, f["cloud"])
        b = fullm.mapping_step_inputs(po["q_w"], po["t_w"], corner, surf, f["cloud"])
        for key in ("q_w", "t_w"):
            assert np.array_equal(a[key], b[key]), key
    for ctx, call in ((reg, lambda c: c.odometry_step()), (odo, lambda c: c.scan_register(scans[0])), (reg, lambda c: c.mapping_enable()),
                      (mp, lambda c: c.odometry_step()), (mp, lambda c: c.cloud(binding.CLOUD_SHARP))):
        with pytest.raises(binding.AloamError) as e:
            call(ctx)
        assert e.value.code == binding.E_STATE
    for c in (full, reg, odo, mp, fullm):
        c.close()


def test_two_hundred_contexts_created_and_destroyed(binding, sequence):
    """Round 5 saw one GPU memory fault in the first aloam_create of a process on one of five fresh boxes and never found a cause.  This hammers
    the path it was on: 200 contexts of every stage mask, mapping enabled where the mask has it (allocation + zero-fill of every buffer + the
    state uploads, then aloam_synchronize, which surfaces device-side faults), a sweep through every 20th, and a handful alive at once."""
    scans, R, t, model = sequence("VLP-16", 2, seed=3, columns=600)
    masks = [binding.STAGE_ALL, binding.STAGE_REGISTRATION, binding.STAGE_ODOMETRY, binding.STAGE_MAPPING, binding.STAGE_REGISTRATION | binding.STAGE_ODOMETRY,
             binding.STAGE_ODOMETRY | binding.STAGE_MAPPING]
    alive = []
    for i in range(200):
        stages = masks[i % len(masks)]
        gpu = binding.Aloam(n_scans=16, min_range=0.3, batch=1 + i % 3, max_points=20000 + 512 * (i % 5), stages=stages)
        if stages & binding.STAGE_MAPPING:
            gpu.mapping_enable(0.2, 0.4, pool_points=4096 << (i % 4))
        gpu.synchronize()
        if stages == binding.STAGE_ALL and i % 20 == 0:
            for x in scans:
                gpu.scan_register([x] * gpu.batch)
                gpu.odometry_step()
                gpu.mapping_step()
            gpu.synchronize()
            assert gpu.map_info()["frame_count"] == 2
        alive.append(gpu)
        if len(alive) > 4:
            alive.pop(0).close()
    for gpu in alive:
        gpu.close()


def test_lm_branch_coverage(O, binding, sequence):
    """The device trust-region loop (lm_device.hpp; reference call src/laserOdometry.cpp:494-499) against the oracle on problems
    built to leave the happy path (tests/lm_scenarios.py): bad warm starts, unobservable directions, fewer residual rows than
    parameters (the device solves the damped normal equations by Cholesky where Ceres and the oracle use QR of the stacked
    Jacobian: this is where the two could part), non-finite residuals.  Every scenario must give the oracle's iteration count,
    success count, termination code and pose; and over the set every branch must actually have been taken."""
    model, scs = lm_scenarios.build(O, sequence)
    seen, worst = set(), 0.0
    gpus = {}
    for sc in scs:
        lm, outer = sc[6], sc[7]
        if (lm, outer) not in gpus:
            gpus[(lm, outer)] = _mk(binding, model, max_points=40000, lm_max_iterations=lm, outer_iterations=outer)
        orc = O.Oracle(n_scans=64, min_range=model.min_range, lm_max_iterations=lm, outer_iterations=outer)
        so, po = lm_scenarios.run(orc, sc)
        sg, pg = lm_scenarios.run(gpus[(lm, outer)], sc)
        for key in ("corner_corr", "plane_corr", "lm_iterations", "lm_successful", "termination"):
            assert so[key] == sg[key], (sc[0], key, so, sg)
        for k in range(2):
            a, b = so["final_cost"][k], sg["final_cost"][k]
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-9 * abs(a) + 1e-16, (sc[0], so, sg)
        _assert_pose_close(po, pg, sc[0])
        worst = max(worst, float(np.abs(po["t_lc"] - pg["t_lc"]).max()))
        seen |= lm_scenarios.branches(so)
    for g in gpus.values():
        g.close()
    assert {"termination0", "termination1", "termination2", "termination3", "termination5", "rejected_or_invalid"} <= seen, seen
    print(f"lm branch coverage: {len(scs)} scenarios, branches {sorted(seen)}, worst |dt| GPU vs oracle {worst:.3g} m")


def test_two_rank_bench_run(binding, tmp_path):
    """`python bench.py --gpus 2` executed for real on this box: the parent starts two ranks (one process each, rendezvous on
    127.0.0.1), which share the single device here (ALOAM_BENCH_SHARED_GPU: gloo control plane, since RCCL refuses two ranks on one
    device), shard 2 x 64 sequences between them without any data-path collective, and rank 0 prints the line.  Its whole-job rate
    must be in the range of a one-rank run over the same 128 sequences; the two lines are kept under gpurun_out/ as evidence."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--frames", "3", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, ALOAM_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r2 = subprocess.run(base + ["--gpus", "2", "--batch", "64"], env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    two = json.loads(r2.stdout.strip().splitlines()[-1])
    r1 = subprocess.run(base + ["--gpus", "1", "--batch", "128"], env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["config"]["sequences_per_gpu"] == 64
    assert two["scaling"] == "weak" and "no collectives" in two["config"]["parallelism"]
    assert 0.5 < two["value"] / one["value"] < 2.0, (two["value"], one["value"])
    out_dir = os.path.join(root, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "two_rank_shared_gpu.json"), "w") as f:
            json.dump({"how": "ALOAM_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --batch 64 --steps 4 --warmup 2 --frames 3 --no-cpu-baseline --no-extras (two ranks sharing ONE MI355X, gloo control plane) next to --gpus 1 --batch 128",
                       "two_ranks": two, "one_rank": one}, f, indent=1)


def test_odometry_step_graph_replay_matches_separate_launches(binding, sequence, monkeypatch):
    """With ALOAM_GRAPH_MAX_BATCH >= batch a context replays the ~15 dependent launches of aloam_odometry_step as one hipGraph launch per
    buffer parity (tried for the ROS shims' batch 1; measured no gain, so off by default).  Same kernels, same arguments: every output must
    equal, bit for bit, what separate launches compute — over enough sweeps to replay both parities several times."""
    scans, R, t, model = sequence("HDL-64", 6, seed=31, columns=512)
    runs = []
    for env in ("8", "0"):
        monkeypatch.setenv("ALOAM_GRAPH_MAX_BATCH", env)
        gpu = _mk(binding, model, batch=2, max_points=max(len(x) for x in scans) + 64)
        rec = []
        for x in scans:
            gpu.scan_register([x, x])
            gpu.odometry_step()
            for b in (0, 1):
                p = gpu.pose(b)
                c = gpu.correspondences(b)
                rec.append((np.concatenate([p["q_w"], p["t_w"], p["q_lc"], p["t_lc"]]), c[0].copy(), c[1].copy(), gpu.cloud(binding.CLOUD_SURF_LAST, b)))
        gpu.close()
        runs.append(rec)
    assert len(runs[0]) == len(runs[1]) == 12
    for a, b in zip(*runs):
        for x, y in zip(a, b):
            assert bits_equal(x, y)


def test_eight_rank_bench_run(binding):
    """The driver's N = 8 launch shape, as far as a one-GPU box allows: `python bench.py --gpus 8` starts eight ranks (one process each,
    rendezvous on 127.0.0.1) that share the single device (ALOAM_BENCH_SHARED_GPU: gloo control plane), 16 sequences each, no data-path
    collective; rank 0 prints n_gpus = 8.  From the memory figures of the line and of a one-rank run: a rank's HBM footprint at the
    default --batch 2048 and eight ranks' host memory must fit a real 8 x MI355X node (288 GB per GPU; the host's RAM)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--frames", "3", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, ALOAM_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r8 = subprocess.run(base + ["--gpus", "8", "--batch", "16"], env=env, capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stderr[-2000:]
    eight = json.loads(r8.stdout.strip().splitlines()[-1])
    assert eight["n_gpus"] == 8 and eight["config"]["sequences_per_gpu"] == 16 and eight["scaling"] == "weak"
    assert "8 x independent sequence shards, no collectives" in eight["config"]["parallelism"]
    r1 = subprocess.run(base + ["--gpus", "1", "--batch", "128"], env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    mem = one["memory"]
    per_seq = mem["hbm_bytes_per_sequence"]
    assert 5e6 < per_seq < 60e6, mem                                            # ~14 MB of state + 3 x 2.1 MB of stored sweeps
    assert per_seq * 2048 + 4e9 < 288e9, mem                                    # a rank at the default batch (2048) on its own 288 GB GPU
    if eight["memory"]["host_rss_bytes"]:
        assert 8 * eight["memory"]["host_rss_bytes"] < 0.8 * eight["memory"]["host_ram_bytes"], eight["memory"]
    out_dir = os.path.join(root, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "eight_rank_shared_gpu.json"), "w") as f:
            json.dump({"how": "ALOAM_BENCH_SHARED_GPU=1 python bench.py --gpus 8 --batch 16 --steps 4 --warmup 2 --frames 3 --no-cpu-baseline --no-extras (eight ranks sharing ONE MI355X, "
                              "gloo control plane: a launch-shape and memory check, not a scaling number) next to --gpus 1 --batch 128",
                       "eight_ranks": eight, "one_rank": one}, f, indent=1)


def test_ring_count_lookback_survives_concurrent_streams():
    """k_ring_features workgroups wait for the counts of the rings in front of them (bounded spin, kErrInternal on time-out).  Four
    contexts on four streams with mapping enabled interleave their launches on the device for 200 steps: the run must finish, and
    aloam_synchronize (called by bench.py at the end of the timed region) must report neither the time-out nor a capacity error
    (the map of a sequence grows by ~1000 points per step here — the refined poses of the replayed sweeps drift by millimetres, so
    the same surfaces land in new voxels — hence the 512 k pool)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "64", "--contexts", "4", "--mapping", "--map-pool", "524288", "--steps", "200",
                        "--warmup", "2", "--frames", "4", "--no-cpu-baseline", "--no-extras", "--repeat-to-seconds", "0"], capture_output=True, text=True, timeout=900)   # no repeated blocks: 200 steps are what the 512 k pool is sized for
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["steps"] == 200 and line["config"]["contexts_per_gpu"] == 4 and line["value"] > 1000


def test_two_ranks_shard_sequences_on_the_hip_path(binding, syn, tmp_path):
    """The N > 1 layout with the HIP path underneath: two processes (gloo control plane, both on this box's one GPU), sequence s on rank
    s mod 2, each rank with its own context over ITS sequences only; the gathered pose table must equal, bit for bit, what one context
    over all five sequences computes — sequences never interact, whichever process or batch slot they run in."""
    import socket
    import subprocess
    import sys
    n_seq, frames = 5, 3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "gpu_shard_worker.py"), str(n_seq), str(frames), str(tmp_path)], env=env))
    assert all(p.wait(timeout=600) == 0 for p in procs)
    t0, t1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(t0, t1)                                            # every rank holds the same gathered table
    seqs = [syn.make_sequence("VLP-16", frames, seed=70 + g, columns=600) for g in range(n_seq)]
    model = seqs[0][3]
    gpu = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, batch=n_seq, max_points=16 * 600 + 64)
    for k in range(frames):
        gpu.scan_register([s[0][k].numpy() for s in seqs])
        gpu.odometry_step()
    gpu.synchronize()
    ref = np.array([np.r_[gpu.pose(b)["t_w"], gpu.pose(b)["q_w"]] for b in range(n_seq)])
    gpu.close()
    assert np.array_equal(t0, ref), np.abs(t0 - ref).max()
    assert np.all(np.linalg.norm(ref[:, :3], axis=1) > 0.5)                  # and every sequence moved


def test_driver_launch_form_with_the_rccl_control_plane(tmp_path):
    """The exact launch form the driver uses for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` — with N = 1, the only N this box has, and the process group forced on
    (ALOAM_BENCH_FORCE_DIST): `nccl` (= RCCL) initialisation on the rank's device, the barriers around the timed region and the MAX
    all-reduce of the elapsed time all execute.  What stays unexecuted here is only RCCL between two devices."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ALOAM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "64", "--steps", "4", "--warmup", "2", "--frames", "3", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["value"] > 1000
    out_dir = os.path.join(root, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "launcher_rccl_one_rank.json"), "w") as f:
            json.dump({"how": "ALOAM_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P bench.py --gpus 1 --batch 64 "
                              "--steps 4 --warmup 2 --frames 3 --no-cpu-baseline --no-extras (nccl = RCCL process group of one rank: init, barrier, MAX all-reduce)", "line": line}, f, indent=1)
