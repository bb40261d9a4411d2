"""GPU parity of the scan-to-map refinement (reference src/laserMapping.cpp) through the C ABI.

Teacher-forced: every frame gets exactly what the mapping node receives (odometry pose, corner_last, surf_last, full
cloud), taken from the committed fixtures that the reference's own translation units produced (tests/golden/refmap_*.npz).
  * vs the oracle in canonical order (same voxel summation order as the HIP path): both down-sampled stacks bit-exact, poses
    to 1e-9.  The refined pose differs from the oracle's in its last bits (parallel f64 reduction + Cholesky on the normal
    equations vs sequential sums + QR), and every inserted map point is q * p + t rounded to f32, so a coordinate that sits on
    a rounding boundary (|z| ~ 1e-5 on flat ground) may land 1 ulp away: free-running maps are compared cube by cube with
    equal point counts and <= 2 ulp per coordinate.  With the solver switched off (lm_max_iterations = 0) the pose is a pure
    composition of inputs and the WHOLE map machinery — window shifts, stable insertion, cube growth, re-filtering — is
    bit-exact (test_mapping_window_shift_and_growth).
  * vs the reference's code itself: poses within 1e-4 m / 1e-4 rad, same occupied cubes, point counts within two per frame and class (measured: <= 1).
"""
import glob
import os

import numpy as np
import pytest
from conftest import bits_equal, quat_angle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAP_GOLDENS = sorted(glob.glob(os.path.join(GOLDEN, "refmap_*.npz")))


def _frames(g):
    for k in range(int(g["frames"])):
        yield k, g[f"odom_q{k}"], g[f"odom_t{k}"], g[f"corner_last{k}"], g[f"surf_last{k}"], g[f"full{k}"]


def _compare_maps(got, want, ctx, exact=True):
    assert set(got) == set(want), (ctx, sorted(set(got) ^ set(want)))
    for cube in want:
        if exact:
            assert bits_equal(got[cube], want[cube]), (ctx, cube, got[cube].shape, want[cube].shape)
        else:
            assert got[cube].shape == want[cube].shape, (ctx, cube, got[cube].shape, want[cube].shape)
            tol = 2 * np.spacing(np.maximum(np.abs(got[cube]), np.abs(want[cube])).astype(np.float32)).astype(np.float64) + 1e-12
            assert np.all(np.abs(got[cube].astype(np.float64) - want[cube]) <= tol), (ctx, cube)


@pytest.mark.parametrize("path", MAP_GOLDENS)
def test_mapping_matches_oracle_and_reference_code(O, binding, path):
    g = np.load(path)
    n_scans, line_res, plane_res = int(g["n_scans"]), float(g["line_res"]), float(g["plane_res"])
    orc = O.Oracle(n_scans=n_scans, min_range=float(g["min_range"]))
    orc.map_config(line_res, plane_res)
    gpu = binding.Aloam(n_scans=n_scans, min_range=float(g["min_range"]), max_points=40000)
    gpu.mapping_enable(line_res, plane_res, pool_points=65536)
    for k, q, t, c, s, f in _frames(g):
        po = orc.mapping_step(q, t, c, s, f)
        pg = gpu.mapping_step_inputs(q, t, c, s, f)
        gpu.synchronize()
        io, ig = orc.map_info(), gpu.map_info()
        for key in ("cenW", "cenH", "cenD", "from_map_corner", "from_map_surf", "corner_stack", "surf_stack", "corner_num0", "corner_num1", "surf_num0", "surf_num1"):
            assert io[key] == ig[key], (path, k, key, io, ig)
        assert bits_equal(orc.map_cloud(O.MAP_CORNER_STACK), gpu.map_cloud(binding.MAP_CORNER_STACK)), (path, k)
        assert bits_equal(orc.map_cloud(O.MAP_SURF_STACK), gpu.map_cloud(binding.MAP_SURF_STACK)), (path, k)
        for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
            assert np.abs(po[key] - pg[key]).max() < 1e-9, (path, k, key, po[key], pg[key])
        for cls in (0, 1):
            _compare_maps(gpu.map_cubes(cls), orc.map_cubes(cls), (path, k, cls), exact=False)
        ro, rg = orc.map_cloud(O.MAP_REGISTERED), gpu.map_cloud(binding.MAP_REGISTERED)
        assert ro.shape == rg.shape and np.abs(ro - rg).max() <= 2e-5
        # the reference's own code (literal std::sort order inside pcl::VoxelGrid): tolerance-level agreement
        assert np.abs(pg["t_w"] - g[f"t_w{k}"]).max() < 1e-4 and quat_angle(pg["q_w"], g[f"q_w{k}"]) < 1e-4
        for cls, name in ((0, "corner_map"), (1, "surf_map")):
            ids, cnt = g[f"{name}_ids{k}"], g[f"{name}_cnt{k}"]
            got = gpu.map_cubes(cls)
            assert set(int(i) for i in ids) == set(got)
            assert abs(int(cnt.sum()) - sum(len(v) for v in got.values())) <= 2      # measured: <= 1 per frame and class
    gpu.close()


@pytest.mark.parametrize("lm_iters", [0, 4])
def test_mapping_window_shift_and_growth(O, binding, lm_iters):
    """Poses that cross several 50 m cubes (window shifts, :323-507), cubes that outgrow their segment and a sort tile, a
    second sequence in the batch that stays put — GPU vs oracle.  lm_iters = 0: solver off, bit-exact maps; 4: free-running."""
    rng = np.random.default_rng(3)
    orc = O.Oracle(16, 0.3, lm_max_iterations=lm_iters)
    orc.map_config(0.4, 0.8)
    orc2 = O.Oracle(16, 0.3, lm_max_iterations=lm_iters)
    orc2.map_config(0.4, 0.8)
    gpu = binding.Aloam(n_scans=16, min_range=0.3, batch=2, max_points=8192, lm_max_iterations=lm_iters)
    gpu.mapping_enable(0.4, 0.8, pool_points=131072)
    track = [(0, 0, 0), (120, -60, 0), (390, -380, 30), (420, -100, 160), (200, 30, 170), (-40, 390, -120), (-380, 395, -130), (-395, 0, 0), (-100, 0, 0)]
    for k, (tx, ty, tz) in enumerate(track):
        pts = rng.uniform(-60, 60, (1500, 4)).astype(np.float32); pts[:, 3] = rng.integers(0, 16, 1500)
        surf = rng.uniform(-60, 60, (4000, 4)).astype(np.float32); surf[:, 2] *= 0.05; surf[:, 3] = rng.integers(0, 16, 4000)
        q = np.array([0, 0, np.sin(0.1 * k), np.cos(0.1 * k)]); t = np.array([tx, ty, tz], float)
        po = orc.mapping_step(q, t, pts, surf, surf[:100])
        po2 = orc2.mapping_step(np.array([0, 0, 0, 1.0]), np.array([0.3 * k, 0, 0]), pts, surf, surf[:50])
        gpu.set_last(pts, surf, 0); gpu.set_full_cloud(surf[:100], 0); gpu.set_state([0, 0, 0, 1], [0, 0, 0], q, t, 0)
        gpu.set_last(pts, surf, 1); gpu.set_full_cloud(surf[:50], 1); gpu.set_state([0, 0, 0, 1], [0, 0, 0], [0, 0, 0, 1.0], [0.3 * k, 0, 0], 1)
        gpu.mapping_step()
        gpu.synchronize()
        for seq, o, p in ((0, orc, po), (1, orc2, po2)):
            pg = gpu.map_pose(seq)
            io, ig = o.map_info(), gpu.map_info(seq)
            assert (io["cenW"], io["cenH"], io["cenD"]) == (ig["cenW"], ig["cenH"], ig["cenD"]), (k, seq, io, ig)
            for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
                assert np.abs(p[key] - pg[key]).max() < 1e-9, (k, seq, key)
            for cls in (0, 1):
                _compare_maps(gpu.map_cubes(cls, seq), o.map_cubes(cls), (k, seq, cls), exact=lm_iters == 0)
    assert gpu.map_info(0)["cenW"] != 10 and max(len(v) for v in gpu.map_cubes(1, 1).values()) > 2048
    gpu.close()


@pytest.mark.parametrize("name,frames,kw,res", [("VLP-16", 5, {"columns": 900}, (0.2, 0.4)), ("HDL-64", 4, {"columns": 1024}, (0.4, 0.8))])
def test_full_pipeline_registration_odometry_mapping(O, binding, sequence, name, frames, kw, res):
    """Raw sweeps in, refined map poses out: scan registration -> odometry -> mapping chained inside one context (nothing
    leaves HBM between the stages), against the oracle chained the same way."""
    scans, R, t, model = sequence(name, frames, seed=31, **kw)
    orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range)
    orc.map_config(*res)
    gpu = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, max_points=max(len(s) for s in scans) + 64)
    gpu.mapping_enable(*res, pool_points=131072)
    for k, x in enumerate(scans):
        orc.scan_register(x)
        po = orc.odometry_step()
        pm = orc.mapping_step(po["q_w"], po["t_w"], orc.cloud(O.CLOUD_CORNER_LAST), orc.cloud(O.CLOUD_SURF_LAST), orc.cloud(O.CLOUD_FULL))
        gpu.scan_register(x)
        gpu.odometry_step()
        gpu.mapping_step()
        gpu.synchronize()
        pg, mg = gpu.pose(), gpu.map_pose()
        assert np.abs(po["t_w"] - pg["t_w"]).max() < 1e-9
        for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
            assert np.abs(pm[key] - mg[key]).max() < 1e-8, (name, k, key, pm[key], mg[key])
        io, ig = orc.map_info(), gpu.map_info()
        for key in ("from_map_corner", "from_map_surf", "corner_stack", "surf_stack", "corner_num1", "surf_num1"):
            assert io[key] == ig[key], (name, k, key, io, ig)
        for cls in (0, 1):
            _compare_maps(gpu.map_cubes(cls), orc.map_cubes(cls), (name, k, cls), exact=False)
    # the refinement pulls the pose towards the synthetic ground truth at least as well as the odometry alone
    gt = R[0].T @ (t[frames - 1] - t[0])
    assert np.linalg.norm(mg["t_w"] - gt) <= np.linalg.norm(pg["t_w"] - gt) + 0.05
    gpu.close()


def test_full_size_mapping_batch(O, binding, syn):
    """BASELINE configs[2] size: HDL-64, 131072 points per sweep, registration -> odometry -> mapping for a batch through the
    device-resident entry point.  Every sequence against its own oracle run; size-independent properties for every sequence: the submap seen
    by frame k is the map left by frame k-1, every down-sampled stack point lands in exactly one cube before the re-filter
    (point-count conservation), the refined trajectory stays within a few centimetres of the synthetic ground truth."""
    import torch
    B, T = 3, 4
    dev = torch.device("cuda", 0)
    model = syn.sensor_model("HDL-64", device=dev)
    NP = model.dirs.shape[0]
    data = torch.zeros((B, T, NP, 4), dtype=torch.float32, device=dev)
    counts = np.zeros((B, T), np.int32)
    world = syn.make_world(555).to(dev)
    gts = []
    for b in range(B):
        R, t = syn.trajectory(T, seed=70 + b, start_angle=0.4 * b)
        gts.append((R.numpy(), t.numpy()))
        gen = torch.Generator(device=dev).manual_seed(70 + b)
        for k in range(T):
            s = syn.render_scan(world, model, R[k], t[k], 0.02, gen)
            counts[b, k] = len(s); data[b, k, :len(s)] = s
    torch.cuda.synchronize()
    gpu = binding.Aloam(n_scans=64, min_range=model.min_range, batch=B, max_points=NP, max_ring_points=2059)
    gpu.mapping_enable(0.4, 0.8, pool_points=262144)
    orcs = [O.Oracle(n_scans=64, min_range=model.min_range) for _ in range(B)]
    for orc in orcs:
        orc.map_config(0.4, 0.8)
    host = data.cpu().numpy()
    prev_total = [0] * B
    for k in range(T):
        gpu.process_device(data.data_ptr() + k * NP * 16, T * NP * 16, counts[:, k])
        gpu.mapping_step()
        gpu.synchronize()
        for b, orc in enumerate(orcs):                                      # every sequence against its own oracle run
            orc.scan_register(host[b, k, :counts[b, k]])
            po = orc.odometry_step()
            pm = orc.mapping_step(po["q_w"], po["t_w"], orc.cloud(O.CLOUD_CORNER_LAST), orc.cloud(O.CLOUD_SURF_LAST), orc.cloud(O.CLOUD_FULL))
            mg = gpu.map_pose(b)
            for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
                assert np.abs(pm[key] - mg[key]).max() < 1e-8, (b, k, key)
            for cls in (0, 1):
                _compare_maps(gpu.map_cubes(cls, b), orc.map_cubes(cls), (b, k, cls), exact=False)
        for b in range(B):
            info = gpu.map_info(b)
            total = sum(len(v) for v in gpu.map_cubes(0, b).values()) + sum(len(v) for v in gpu.map_cubes(1, b).values())
            assert info["from_map_corner"] + info["from_map_surf"] == prev_total[b], (k, b, info, prev_total[b])   # the window holds the whole map here
            assert prev_total[b] < total <= prev_total[b] + info["corner_stack"] + info["surf_stack"]
            assert info["corner_stack"] > 1000 and info["surf_stack"] > 5000 and info["frame_count"] == k + 1
            if k > 0:
                assert info["corner_num1"] > 500 and info["surf_num1"] > 2000
            prev_total[b] = total
    for b in range(B):
        Rg, tg = gts[b]
        gt = Rg[0].T @ (tg[T - 1] - tg[0])
        assert np.linalg.norm(gpu.map_pose(b)["t_w"] - gt) < 0.06, (b, gpu.map_pose(b)["t_w"], gt)
    gpu.close()


def test_map_pool_compaction(O, binding):
    """A map that keeps growing inside one cube outgrows its segment again and again; the abandoned segments would exhaust a
    small pool.  The automatic compaction must keep the run going and leave the map bit-exact (solver off)."""
    rng = np.random.default_rng(11)
    orc = O.Oracle(16, 0.3, lm_max_iterations=0)
    orc.map_config(0.4, 0.8)
    gpu = binding.Aloam(n_scans=16, min_range=0.3, batch=1, max_points=8192, lm_max_iterations=0)
    gpu.mapping_enable(0.4, 0.8, pool_points=40960, pool_limit=40960)        # a pool that may not grow: compaction has to do it
    for k in range(18):
        corner = rng.uniform(-10, 10, (1500, 4)).astype(np.float32); corner[:, 3] = rng.integers(0, 16, 1500)
        surf = rng.uniform(-20, 20, (3000, 4)).astype(np.float32); surf[:, 2] *= 0.01; surf[:, 3] = rng.integers(0, 16, 3000)
        q, t = np.array([0, 0, 0, 1.0]), np.array([0.01 * k, 0, 0])
        orc.mapping_step(q, t, corner, surf, surf[:10])
        gpu.set_last(corner, surf, 0); gpu.set_full_cloud(surf[:10], 0); gpu.set_state([0, 0, 0, 1], [0, 0, 0], q, t, 0)
        gpu.mapping_step()
        gpu.synchronize()                                     # raises ALOAM_E_CAPACITY if the pool ran out
        for cls in (0, 1):
            _compare_maps(gpu.map_cubes(cls), orc.map_cubes(cls), (k, cls))
    info = gpu.map_info()
    assert info["compactions"] >= 1, info
    assert sum(len(v) for v in gpu.map_cubes(0).values()) > 20480            # more live points than half the pool: doubling alone could not have held them
    gpu.close()


def test_map_pool_growth_keeps_the_map_bit_exact(O, binding):
    """The reference's cubes are std::vectors (src/laserMapping.cpp:737-783): a map grows as long as points arrive.  A pool that starts at 8192
    points per class must double again and again under the same load as above - no point dropped (aloam_synchronize would raise), the whole
    map bit-exact against the oracle after every frame (solver off), the growths reported."""
    rng = np.random.default_rng(12)
    orc = O.Oracle(16, 0.3, lm_max_iterations=0)
    orc.map_config(0.4, 0.8)
    gpu = binding.Aloam(n_scans=16, min_range=0.3, batch=1, max_points=8192, lm_max_iterations=0)
    gpu.mapping_enable(0.4, 0.8, pool_points=8192)
    pools = []
    for k in range(14):
        corner = rng.uniform(-10 - 2 * k, 10 + 2 * k, (1500, 4)).astype(np.float32); corner[:, 3] = rng.integers(0, 16, 1500)
        surf = rng.uniform(-20 - 4 * k, 20 + 4 * k, (3000, 4)).astype(np.float32); surf[:, 2] *= 0.01; surf[:, 3] = rng.integers(0, 16, 3000)
        q, t = np.array([0, 0, 0, 1.0]), np.array([0.01 * k, 0, 0])
        orc.mapping_step(q, t, corner, surf, surf[:10])
        gpu.set_last(corner, surf, 0); gpu.set_full_cloud(surf[:10], 0); gpu.set_state([0, 0, 0, 1], [0, 0, 0], q, t, 0)
        gpu.mapping_step()
        gpu.synchronize()
        for cls in (0, 1):
            _compare_maps(gpu.map_cubes(cls), orc.map_cubes(cls), (k, cls))
        pools.append(gpu.map_pool_info())
    live = max(sum(len(v) for v in gpu.map_cubes(cls).values()) for cls in (0, 1))
    assert pools[-1]["growths"] >= 2 and pools[-1]["pool_points"] >= 32768 and pools[-1]["live_max"] == live, (pools[-1], live)
    assert all(p["live_max"] <= p["pool_points"] for p in pools)
    gpu.close()


def test_capacity_errors_of_unsynchronised_steps_are_not_lost(binding):
    """Asynchronous use (bench.py queues many steps and synchronises once): a step that runs out of map pool is followed by steps
    that fit.  The per-step flag is cleared by the next step, but aloam_synchronize must still report ALOAM_E_CAPACITY — once —
    and the map must hold the points of the steps that did fit."""
    rng = np.random.default_rng(5)
    gpu = binding.Aloam(n_scans=16, min_range=0.3, batch=2, max_points=16384, lm_max_iterations=0)
    gpu.mapping_enable(0.4, 0.8, pool_points=4096, pool_limit=4096)            # at its ceiling from the start

    def frame(n_surf, seq_with_points, half):
        for b in range(2):
            n = n_surf if b == seq_with_points else 0
            surf = rng.uniform(-half, half, (n, 4)).astype(np.float32); surf[:, 2] *= 0.01; surf[:, 3] = rng.integers(0, 16, n)
            corner = rng.uniform(-5, 5, (min(n, 20), 4)).astype(np.float32); corner[:, 3] = rng.integers(0, 16, min(n, 20))
            gpu.set_last(corner, surf, b); gpu.set_full_cloud(surf[:4], b); gpu.set_state([0, 0, 0, 1], [0, 0, 0], [0, 0, 0, 1.0], [0, 0, 0.0], b)
        gpu.mapping_step()

    frame(200, 1, 20.0)         # fits
    gpu.synchronize()
    held = sum(len(v) for v in gpu.map_cubes(1, 1).values())
    assert held > 100
    frame(14000, 1, 40.0)       # sequence 1: ~7000 occupied 0.8 m voxels on an 80 m square, more than the 4096-point pool holds even packed -> dropped
    frame(0, 1, 20.0)           # nothing to insert: this step clears the per-step flag
    frame(150, 0, 20.0)         # sequence 0 fits
    with pytest.raises(binding.AloamError) as e:
        gpu.synchronize()
    assert e.value.code == binding.E_CAPACITY and "sequence 1" in str(e.value)
    gpu.synchronize()           # reported once
    assert sum(len(v) for v in gpu.map_cubes(1, 1).values()) >= held       # the earlier map is intact
    assert sum(len(v) for v in gpu.map_cubes(1, 0).values()) > 50           # and the later, fitting step was inserted
    gpu.close()


@pytest.mark.parametrize("case", ["many_runs", "far_point", "voxel_overflow", "big_segment"])
def test_voxel_filter_segments_the_lds_kernel_hands_to_the_general_path(O, binding, case):
    """k_vox_lds takes a segment only when its runs fit the LDS, its coordinates keep floor(p / leaf) an exact f32 integer below 2^23
    and it has at most 65535 points; everything else must come out of the tile-sort / rank-merge path of round 2 — unchanged results:
    the down-sampled incoming clouds (laserCloudCornerStack / SurfStack, reference src/laserMapping.cpp:542-550) and the re-filtered
    cubes (:788-801) bit for bit against the oracle, solver off."""
    rng = np.random.default_rng(17)
    cap = 90000
    orc = O.Oracle(16, 0.3, lm_max_iterations=0)
    orc.map_config(0.4, 0.8)
    gpu = binding.Aloam(n_scans=16, min_range=0.3, batch=1, max_points=cap, lm_max_iterations=0)
    gpu.mapping_enable(0.4, 0.8, pool_points=262144)
    corner = rng.uniform(-30, 30, (1500, 4)).astype(np.float32); corner[:, 3] = np.sort(rng.integers(0, 16, 1500))
    if case == "many_runs":            # 30 000 scattered points: every point its own run, more than the 24 576 the LDS holds
        surf = rng.uniform(-60, 60, (30000, 4)).astype(np.float32); surf[:, 2] *= 0.05
    elif case == "far_point":          # one coordinate beyond 2^23 leaves: the exact-integer argument of the cell test does not hold
        surf = rng.uniform(-20, 20, (5000, 4)).astype(np.float32); surf[:, 2] *= 0.05
        surf[1234, 0] = 7.5e6
    elif case == "voxel_overflow":     # more than 2^31 voxels in the box: pcl::VoxelGrid returns its input unfiltered
        surf = rng.uniform(-20, 20, (3000, 4)).astype(np.float32)
        surf[7, :3] = (3.0e5, 3.0e5, 3.0e5); surf[8, :3] = (-3.0e5, -3.0e5, -3.0e5)   # 7.5e5 leaves per axis: 4e17 voxels (no 64-bit wrap)
    else:                              # more points than the 16-bit point index of a run key addresses
        surf = rng.uniform(-40, 40, (80000, 4)).astype(np.float32); surf[:, 2] *= 0.02
    surf[:, 3] = np.sort(rng.integers(0, 16, len(surf)))
    for k in range(2):                 # two frames: the second one re-filters cubes that already hold points
        q, t = np.array([0, 0, 0, 1.0]), np.array([0.05 * k, 0, 0])
        orc.mapping_step(q, t, corner, surf, surf[:10])
        gpu.set_last(corner, surf, 0); gpu.set_full_cloud(surf[:10], 0); gpu.set_state([0, 0, 0, 1], [0, 0, 0], q, t, 0)
        gpu.mapping_step()
        gpu.synchronize()
        so, sg = orc.map_cloud(O.MAP_SURF_STACK), gpu.map_cloud(binding.MAP_SURF_STACK)
        assert so.shape == sg.shape and bits_equal(so, sg), (case, k, so.shape, sg.shape)
        assert bits_equal(orc.map_cloud(O.MAP_CORNER_STACK), gpu.map_cloud(binding.MAP_CORNER_STACK)), (case, k)
        for cls in (0, 1):
            _compare_maps(gpu.map_cubes(cls), orc.map_cubes(cls), (case, k, cls))
    gpu.close()


def test_pools_grow_under_asynchronous_steps_without_dropping_a_point(binding, syn):
    """The throughput use: sweeps resident on the device, aloam_process_device + aloam_mapping_step queued for 120 frames WITHOUT a single synchronisation
    in between, four travelling sequences, pools that start far too small (16 384 points).  The host has to size the pools ahead of the device from
    the occupancy report of steps that finished a few steps ago (map_ensure_capacity): no ALOAM_E_CAPACITY at the end, the pools have grown, and the
    map, the window and the poses are those of a run that synchronises after every step with pools that never need to grow - bit for bit."""
    import torch
    B, T, cols = 4, 120, 512
    dev = torch.device("cuda", 0)
    model = syn.sensor_model("HDL-64", columns=cols, device=dev)
    NP = model.dirs.shape[0]
    data = torch.zeros((B, T, NP, 4), dtype=torch.float32, device=dev)
    counts = np.zeros((B, T), np.int32)
    for b in range(B):
        world = syn.make_street_world(70 + b).to(dev)
        R, t = syn.trajectory_travel(T, step=1.6, seed=70 + b)
        gen = torch.Generator(device=dev).manual_seed(500 + b)
        for k in range(T):
            s = syn.render_scan(world, model, R[k], t[k], 0.02, gen, max_range=syn.STREET_MAX_RANGE, cull=True)
            counts[b, k] = s.shape[0]
            data[b, k, : s.shape[0]] = s
    torch.cuda.synchronize()
    results = []
    for asynchronous in (True, False):
        gpu = binding.Aloam(n_scans=64, min_range=5.0, batch=B, max_points=NP)
        gpu.mapping_enable(0.4, 0.8, pool_points=16384 if asynchronous else 524288)
        for k in range(T):
            gpu.process_device(data.data_ptr() + k * NP * 16, T * NP * 16, counts[:, k])
            gpu.mapping_step()
            if not asynchronous:
                gpu.synchronize()
        gpu.synchronize()                                      # raises ALOAM_E_CAPACITY if any step dropped points
        pool = gpu.map_pool_info()
        results.append(([gpu.map_pose(b) for b in range(B)], [[gpu.map_cubes(cls, b) for cls in (0, 1)] for b in range(B)], [gpu.map_info(b) for b in range(B)], pool))
        gpu.close()
    (pa, ca, ia, pool_a), (ps, cs, is_, pool_s) = results
    assert pool_a["growths"] >= 2 and pool_a["pool_points"] >= pool_a["live_max"] > 16384 and pool_s["growths"] == 0, (pool_a, pool_s)
    for b in range(B):
        for key in ("q_w", "t_w", "q_wmap_wodom", "t_wmap_wodom"):
            assert np.array_equal(pa[b][key], ps[b][key]), (b, key)
        assert {k: v for k, v in ia[b].items() if k != "compactions"} == {k: v for k, v in is_[b].items() if k != "compactions"}, (b, ia[b], is_[b])
        for cls in (0, 1):
            _compare_maps(ca[b][cls], cs[b][cls], (b, cls))
