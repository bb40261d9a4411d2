"""CPU tests of the oracle (oracle/): it is the checker for the HIP path, so it is pinned here against
independent implementations — libm, scipy cKDTree, numpy f32 re-computation, central differences, dual numbers,
scipy minimisers, synthetic ground truth — and against the committed self-generated goldens.

The reference has no tests or golden vectors of its own: parity with the real PCL + Ceres build
stays UNPINNED, and these cross-checks are what stands in for it.
"""
import ctypes
import glob
import os

import numpy as np
import pytest
from conftest import bits_equal, quat_angle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------------------
def test_atan2f_port_is_bit_identical_to_libm(O):
    """The FDLIBM-style atan2f the HIP kernels carry must equal glibc's atan2f (what scanRegistration.cpp:141,208 call)."""
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    libm.atan2f.restype = ctypes.c_float
    rng = np.random.default_rng(0)
    ys = np.concatenate([rng.normal(size=20000) * 30, rng.normal(size=5000) * 1e-3, [0.0, -0.0, 1.0, -1.0, 5.0, 1e-30, 3e38]]).astype(np.float32)
    xs = np.concatenate([rng.normal(size=20000) * 30, rng.normal(size=5000) * 50, [0.0, -0.0, 1.0, -1.0, 0.0, -1e-30, -3e38]]).astype(np.float32)
    n = min(len(xs), len(ys))
    for y, x in zip(ys[:n], xs[:n]):
        a = np.float32(libm.atan2f(float(y), float(x)))
        b = np.float32(O.atan2f_port(y, x))
        assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b)), (y, x, a, b)


def test_nn_search_kdtree_equals_brute_force_and_scipy(O):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    target = (rng.normal(size=(6000, 4)) * [20, 20, 2, 1]).astype(np.float32)
    query = (rng.normal(size=(900, 4)) * [20, 20, 2, 1]).astype(np.float32)
    i_kd, d_kd = O.nn_search(target, query, brute=False)
    i_bf, d_bf = O.nn_search(target, query, brute=True)
    assert np.array_equal(i_kd, i_bf) and bits_equal(d_kd, d_bf)
    _, i_sp = cKDTree(target[:, :3].astype(np.float64)).query(query[:, :3].astype(np.float64))
    assert (i_sp == i_kd).mean() > 0.999          # f64 vs f32 distance can only differ on near-ties
    dd = target[i_kd, :3].astype(np.float64) - query[:, :3].astype(np.float64)
    assert np.allclose((dd ** 2).sum(1), d_kd, rtol=1e-5)
    # ties -> lowest index
    dup = np.vstack([target[:10], target[:10]])
    i_t, _ = O.nn_search(dup, target[:10], brute=False)
    assert np.array_equal(i_t, np.arange(10))


def test_voxel_filter_matches_independent_numpy_model(O):
    rng = np.random.default_rng(2)
    pts = (rng.normal(size=(3000, 4)) * [6, 6, 1, 0.01] + [0, 0, 0, 7.05]).astype(np.float32)
    leaf = np.float32(0.2)
    out = O.voxel_filter(pts, 0.2, canonical=True)
    inv = np.float32(1.0) / leaf
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64)
    ijk -= np.floor(pts[:, :3].min(0) * inv).astype(np.int64)
    div = ijk.max(0) + 1
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq = np.unique(idx)
    assert len(out) == len(uniq)                  # one output per occupied voxel, ascending voxel index
    for k, u in enumerate(uniq[:200]):
        members = pts[idx == u]
        acc = np.zeros(4, np.float32)
        for m in members:                         # f32 accumulation in input order, then / count
            acc = acc + m
        assert bits_equal(out[k], acc / np.float32(len(members)))
    # the reference's std::sort leaves the in-voxel order unspecified: centroids may differ in the last bits only
    out_std = O.voxel_filter(pts, 0.2, canonical=False)
    assert out_std.shape == out.shape and np.abs(out_std - out).max() < 1e-5
    assert len(O.voxel_filter(np.zeros((0, 4), np.float32), 0.2)) == 0


def test_voxel_filter_overflow_guard_returns_input(O):
    pts = np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 2]], np.float32)      # > INT_MAX voxels -> PCL passes the input through
    assert bits_equal(O.voxel_filter(pts, 0.2), pts)


# ---------------------------------------------------------------------------------------------------------
def _random_pose(rng, ang=0.3):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = rng.uniform(-ang, ang)
    return np.r_[np.sin(a / 2) * ax, np.cos(a / 2)], rng.normal(size=3)


@pytest.mark.parametrize("kind,n", [(0, 9), (1, 12)])
def test_factor_jacobians_autodiff_vs_closed_form_vs_central_differences(O, kind, n):
    """Dual-number autodiff of the functors as written (what Ceres evaluates) == closed form the GPU uses == numeric."""
    rng = np.random.default_rng(3 + kind)
    for _ in range(100):
        consts = rng.normal(size=n) * 10
        q, t = _random_pose(rng)
        r_ad, J_ad = O.factor_eval(kind, consts, q, t, analytic=False)
        r_cf, J_cf = O.factor_eval(kind, consts, q, t, analytic=True)
        assert np.allclose(r_ad, r_cf, rtol=1e-12, atol=1e-12)
        assert np.allclose(J_ad, J_cf, rtol=1e-10, atol=1e-10)
        h = 1e-6
        J_fd = np.zeros_like(J_cf)
        for k in range(6):
            d = np.zeros(6); d[k] = h
            rp, _ = O.factor_eval(kind, consts, O.quat_plus(q, d[:3]), t + d[3:], True)
            rm, _ = O.factor_eval(kind, consts, O.quat_plus(q, -d[:3]), t - d[3:], True)
            J_fd[:, k] = (rp - rm) / (2 * h)
        assert np.allclose(J_fd, J_cf, rtol=1e-5, atol=1e-6 * max(1.0, np.abs(J_cf).max()))


@pytest.mark.parametrize("kind,n", [(0, 9), (1, 12)])
def test_factor_jacobians_with_interpolation_ratio(O, kind, n):
    """DISTORTION 1: the functors with a general interpolation ratio s.  s = 1 reproduces the default evaluation bit for bit; for
    other s the dual-number Jacobian equals central differences on the manifold, and an independent numpy restatement of
    lp = slerp(I, q, s) * cp + s t gives the same residual."""
    rng = np.random.default_rng(40 + kind)
    for it in range(60):
        consts = rng.normal(size=n) * 10
        q, t = _random_pose(rng)
        r1, J1 = O.factor_eval_s(kind, consts, 1.0, q, t)
        r0, J0 = O.factor_eval(kind, consts, q, t, analytic=False)
        assert np.array_equal(r1, r0) and np.array_equal(J1, J0)
        s = [0.0, 0.3, 0.999, 1.7, 9.9][it % 5]                              # > 1 happens: intensity = ring - eps
        r, J = O.factor_eval_s(kind, consts, s, q, t)
        # independent restatement of the interpolated transform
        w = np.clip(abs(q[3]), -1, 1)
        if w >= 1.0 - np.finfo(float).eps:
            c0, c1 = 1 - s, s
        else:
            th = np.arccos(w); c0, c1 = np.sin((1 - s) * th) / np.sin(th), np.sin(s * th) / np.sin(th)
        if q[3] < 0:
            c1 = -c1
        u, ww = c1 * q[:3], c0 + c1 * q[3]
        cp = consts[:3]
        uv = 2 * np.cross(u, cp)
        lp = cp + ww * uv + np.cross(u, uv) + s * t
        if kind == 0:
            a, b = consts[3:6], consts[6:9]
            ref = np.cross(lp - a, lp - b) / np.linalg.norm(a - b)
        else:
            j, l, m = consts[3:6], consts[6:9], consts[9:12]
            nrm = np.cross(j - l, j - m); nrm /= np.linalg.norm(nrm)
            ref = np.array([(lp - j) @ nrm])
        assert np.allclose(r, ref, rtol=1e-11, atol=1e-11)
        h = 1e-6
        J_fd = np.zeros_like(J)
        for k in range(6):
            d = np.zeros(6); d[k] = h
            rp, _ = O.factor_eval_s(kind, consts, s, O.quat_plus(q, d[:3]), t + d[3:])
            rm, _ = O.factor_eval_s(kind, consts, s, O.quat_plus(q, -d[:3]), t - d[3:])
            J_fd[:, k] = (rp - rm) / (2 * h)
        assert np.allclose(J_fd, J, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(J).max())), (kind, s, J_fd, J)


def test_edge_residual_is_point_to_line_distance(O):
    rng = np.random.default_rng(5)
    a, b, cp = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3) * 3
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    r, _ = O.factor_eval(0, np.r_[cp, a, b], q, t, True)
    dist = np.linalg.norm(np.cross(cp - a, cp - b)) / np.linalg.norm(a - b)
    assert np.isclose(np.linalg.norm(r), dist)


def _synthetic_problem(rng, n_e=60, n_p=120, noise=0.02, outliers=5):
    """Correspondences generated from a known relative pose (q_true, t_true): p_last = R p_curr + t."""
    from scipy.spatial.transform import Rotation
    q_true, t_true = _random_pose(rng, 0.05)
    t_true = t_true * 0.3
    Rm = Rotation.from_quat(q_true).as_matrix()
    edges, planes = [], []
    for _ in range(n_e):
        cp = rng.normal(size=3) * 10
        lp = Rm @ cp + t_true
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        edges.append(np.r_[cp + rng.normal(size=3) * noise, lp + 0.7 * d, lp - 0.9 * d])
    for _ in range(n_p):
        cp = rng.normal(size=3) * 10
        lp = Rm @ cp + t_true
        u, v = rng.normal(size=3), rng.normal(size=3)
        planes.append(np.r_[cp + rng.normal(size=3) * noise, lp, lp + u, lp + v])
    for k in range(outliers):
        planes[k][:3] += rng.normal(size=3) * 2
    return np.array(edges), np.array(planes), q_true, t_true


def test_lm_many_iterations_reaches_the_minimum_scipy_finds(O):
    """Cost function + derivatives + trust-region loop: run to convergence and compare with an independent BFGS."""
    from scipy.optimize import minimize
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(6)
    edges, planes, q_true, t_true = _synthetic_problem(rng)
    q0, t0 = np.array([0, 0, 0, 1.0]), np.zeros(3)
    q, t, info = O.lm_solve(edges, planes, q0, t0, max_iterations=60)
    assert info["final_cost"] < info["initial_cost"] * 0.2
    assert np.abs(t - t_true).max() < 0.02 and quat_angle(q, q_true) < 2e-3

    def f(x):
        qq = Rotation.from_rotvec(x[:3]).as_quat()
        return O.cost(edges, planes, qq, x[3:])
    res = minimize(f, np.r_[Rotation.from_quat(q).as_rotvec(), t], method="BFGS", options={"gtol": 1e-12})
    assert res.fun <= info["final_cost"] + 1e-12
    assert info["final_cost"] - res.fun < 1e-5 * res.fun      # LM stops on function_tolerance 1e-6 (relative)
    assert np.abs(res.x[3:] - t).max() < 1e-4


def test_lm_four_iterations_semantics(O):
    rng = np.random.default_rng(7)
    edges, planes, *_ = _synthetic_problem(rng)
    q0, t0 = np.array([0, 0, 0, 1.0]), np.zeros(3)
    q, t, info = O.lm_solve(edges, planes, q0, t0, max_iterations=4)
    assert 1 <= info["iterations"] <= 4 and info["successful"] <= info["iterations"]
    assert info["final_cost"] <= info["initial_cost"]
    assert np.isclose(info["initial_cost"], O.cost(edges, planes, q0, t0))
    assert np.isclose(info["final_cost"], O.cost(edges, planes, q, t))
    assert np.isclose(np.linalg.norm(q), 1.0, atol=1e-12)
    # autodiff vs closed-form evaluation give the same iterate to rounding
    q2, t2, _ = O.lm_solve(edges, planes, q0, t0, max_iterations=4, analytic=True)
    assert np.abs(q - q2).max() < 1e-10 and np.abs(t - t2).max() < 1e-10
    # no residuals -> untouched, termination "no residuals"
    q3, t3, info3 = O.lm_solve(np.zeros((0, 9)), np.zeros((0, 12)), q0, t0)
    assert info3["termination"] == 4 and np.array_equal(q3, q0)


def test_huber_is_applied_per_block(O):
    """3-residual edge block: rho acts on the squared norm of the block (Ceres), not per scalar."""
    cp, a, b = np.array([0.0, 1.0, 0.0]), np.array([0.0, 0, 0]), np.array([1.0, 0, 0])   # distance 1 > 0.1
    c = O.cost(np.r_[cp, a, b][None], np.zeros((0, 12)), [0, 0, 0, 1], [0, 0, 0])
    assert np.isclose(c, 0.5 * (2 * 0.1 * 1.0 - 0.01))


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["VLP-16", "HDL-64"])
def test_registration_invariants(O, sequence, name):
    scans, R, t, model = sequence(name, 2, seed=1)
    orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range)
    f = orc.scan_register(scans[0])
    start, count = orc.ring_ranges()
    curv, label, picked = orc.per_point()
    cloud = f["cloud"]
    assert count.sum() == len(cloud) <= len(scans[0])
    assert np.all(np.diff(start) == count[:-1])
    ring_of = np.repeat(np.arange(model.n_scans), count)
    frac = cloud[:, 3] - ring_of
    assert np.all((frac > -0.06) & (frac < 0.16))                     # intensity = ring + 0.1 * relTime
    if name == "HDL-64":
        assert count[51:].sum() == 0                                  # only rings 0..50 are used (scanRegistration.cpp:195)
    r2 = (cloud[:, :3].astype(np.float64) ** 2).sum(1)
    assert r2.min() >= model.min_range ** 2 * (1 - 1e-6)
    # curvature: independent numpy f32 evaluation in the same left-to-right order
    c = cloud[:, :3]
    n = len(c)
    i = np.arange(5, n - 5)
    acc = c[i - 5].copy()
    for off in (-4, -3, -2, -1):
        acc = acc + c[i + off]
    acc = acc - np.float32(10) * c[i]
    for off in (1, 2, 3, 4, 5):
        acc = acc + c[i + off]
    ref = acc[:, 0] * acc[:, 0] + acc[:, 1] * acc[:, 1] + acc[:, 2] * acc[:, 2]
    assert bits_equal(curv[5:n - 5], ref.astype(np.float32))
    # selection rules
    n_sectors = 6 * int((count >= 17).sum())
    assert len(f["sharp"]) <= 2 * n_sectors and len(f["less_sharp"]) <= 20 * n_sectors and len(f["flat"]) <= 4 * n_sectors
    assert (label == 2).sum() == len(f["sharp"]) and (label >= 1).sum() == len(f["less_sharp"]) and (label == -1).sum() == len(f["flat"])
    assert np.all(curv[label >= 1] > 0.1) and np.all(curv[label == -1] < 0.1)
    sharp_set = {p.tobytes() for p in f["sharp"]}
    assert sharp_set <= {p.tobytes() for p in f["less_sharp"]}
    for s0, c0 in zip(start, count):                                  # first / last 5 points of a ring are never selectable
        if c0 >= 17:
            assert not label[s0:s0 + 5].any() and not label[s0 + c0 - 6:s0 + c0].any()
    assert np.all(np.diff(np.floor(f["less_flat"][:, 3] + 0.06)) >= 0)   # less-flat cloud stays ring-sorted


def test_curvature_ties_do_not_change_the_result_on_test_data(O, sequence):
    """std::sort's tie order is unspecified (scanRegistration.cpp:288); on the noisy synthetic data the canonical
    (curvature, index) order and libstdc++'s order select the same features, and the voxel centroids agree to 1e-5."""
    scans, *_ , model = sequence("VLP-16", 2, seed=1)
    a = O.Oracle(n_scans=16, min_range=model.min_range, canonical_order=True)
    b = O.Oracle(n_scans=16, min_range=model.min_range, canonical_order=False)
    fa, fb = a.scan_register(scans[0]), b.scan_register(scans[0])
    for k in ("cloud", "sharp", "less_sharp", "flat"):
        assert bits_equal(fa[k], fb[k])
    assert fa["less_flat"].shape == fb["less_flat"].shape and np.abs(fa["less_flat"] - fb["less_flat"]).max() < 1e-5
    a.odometry_step(); b.odometry_step()
    fa, fb = a.scan_register(scans[1]), b.scan_register(scans[1])
    pa, pb = a.odometry_step(), b.odometry_step()
    assert np.abs(pa["t_lc"] - pb["t_lc"]).max() < 1e-5 and quat_angle(pa["q_lc"], pb["q_lc"]) < 1e-5


def test_registration_edge_cases(O):
    orc = O.Oracle(n_scans=16, min_range=0.3)
    with pytest.raises(RuntimeError):
        orc.scan_register(np.zeros((0, 4), np.float32))
    with pytest.raises(RuntimeError):
        orc.scan_register(np.full((10, 4), np.nan, np.float32))
    with pytest.raises(RuntimeError):                                   # everything inside minimum_range
        orc.scan_register(np.full((10, 4), 0.01, np.float32))
    with pytest.raises(RuntimeError):
        O.Oracle(n_scans=48).scan_register(np.ones((10, 4), np.float32))  # unsupported scan_line without ring_from_field
    f = orc.scan_register(np.array([[5, 0, 0, 0], [5, 1, 0, 0], [5, 2, 0.2, 0]], np.float32))   # rings too short: no features
    assert len(f["cloud"]) == 3 and len(f["sharp"]) == 0 and len(f["less_flat"]) == 0
    # stride 32 (PointCloud2 layout) gives the same result as stride 16
    rng = np.random.default_rng(0)
    pts = (rng.normal(size=(500, 4)) * [10, 10, 0.5, 1]).astype(np.float32)
    wide = np.zeros((500, 8), np.float32); wide[:, :4] = pts
    f16, f32 = O.Oracle(16, 0.3).scan_register(pts), O.Oracle(16, 0.3).scan_register(wide)
    assert bits_equal(f16["cloud"], f32["cloud"])


@pytest.mark.parametrize("name,tol", [("VLP-16", 0.08), ("HDL-64", 0.06)])
def test_odometry_tracks_ground_truth(O, syn, sequence, name, tol):
    import torch
    scans, R, t, model = sequence(name, 5, seed=2)
    orc = O.Oracle(n_scans=model.n_scans, min_range=model.min_range)
    for k, s in enumerate(scans):
        orc.scan_register(s)
        p = orc.odometry_step()
        if k >= 2:                                                    # first pair starts from identity and is not converged
            R_lc, t_lc = syn.relative_pose(torch.from_numpy(R), torch.from_numpy(t), k)
            assert np.abs(p["t_lc"] - t_lc.numpy()).max() < tol
            from scipy.spatial.transform import Rotation
            assert quat_angle(p["q_lc"], Rotation.from_matrix(R_lc.numpy()).as_quat()) < 0.02
    st = orc.odom_stats()
    assert st["corner_corr"][1] > 50 and st["plane_corr"][1] > 200


def test_first_frame_does_not_solve(O, sequence):
    scans, *_, model = sequence("VLP-16", 2, seed=1)
    orc = O.Oracle(16, model.min_range)
    orc.scan_register(scans[0])
    p = orc.odometry_step()
    assert np.array_equal(p["q_w"], [0, 0, 0, 1]) and np.array_equal(p["t_w"], [0, 0, 0])
    assert orc.odom_stats()["lm_iterations"] == [0, 0]
    assert len(orc.cloud(O.CLOUD_CORNER_LAST)) == len(orc.cloud(O.CLOUD_LESS_SHARP))


def test_kdtree_and_brute_force_give_identical_odometry(O, sequence):
    scans, *_, model = sequence("VLP-16", 3, seed=1)
    a, b = O.Oracle(16, model.min_range, nn_brute=False), O.Oracle(16, model.min_range, nn_brute=True)
    for s in scans:
        a.scan_register(s); b.scan_register(s)
        pa, pb = a.odometry_step(), b.odometry_step()
    assert np.array_equal(pa["t_w"], pb["t_w"]) and np.array_equal(pa["q_w"], pb["q_w"])


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("ref")))
def test_oracle_reproduces_committed_goldens(O, path):
    g = np.load(path)
    orc = O.Oracle(n_scans=int(g["n_scans"]), min_range=float(g["min_range"]))
    k = 0
    while f"scan{k}" in g:
        f = orc.scan_register(g[f"scan{k}"])
        p = orc.odometry_step()
        for key in ("sharp", "less_sharp", "flat", "less_flat"):
            assert bits_equal(f[key], g[f"{key}{k}"]), (path, k, key)
        assert bits_equal(f["cloud"][:, 3], g[f"cloud_intensity{k}"])
        for key in ("q_lc", "t_lc", "q_w", "t_w"):
            assert np.allclose(p[key], g[f"{key}{k}"], rtol=0, atol=1e-9), (path, k, key)
        k += 1
    assert k >= 2


# ---------------------------------------------------------------------------------------------------------
# third-party stand-ins of the mapping stage: KdTreeFLANN k = 5, SelfAdjointEigenSolver, colPivHouseholderQr
def test_knn5_equals_brute_force_and_scipy(O):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(70)
    tgt = np.zeros((6000, 4), np.float32); tgt[:, :3] = rng.uniform(-20, 20, (6000, 3))
    qry = np.zeros((500, 4), np.float32); qry[:, :3] = rng.uniform(-22, 22, (500, 3))
    i_tree, d_tree = O.knn_search(tgt, qry, 5)
    i_bf, d_bf = O.knn_search(tgt, qry, 5, brute=True)
    assert np.array_equal(i_tree, i_bf) and bits_equal(d_tree, d_bf)
    assert np.all(np.diff(d_tree, axis=1) >= 0)                                           # ascending, like nearestKSearch
    _, i_sp = cKDTree(tgt[:, :3].astype(np.float64)).query(qry[:, :3].astype(np.float64), k=5)
    same = (i_sp == i_tree).all(axis=1)
    assert same.mean() > 0.99                                                             # f32 vs f64 distances may swap near-ties
    for r in np.where(~same)[0]:
        assert set(i_sp[r]) == set(i_tree[r]) or np.allclose(np.sort(d_tree[r]), np.sort(((tgt[i_sp[r], :3] - qry[r, :3]) ** 2).sum(1)), rtol=1e-5)
    few = O.knn_search(tgt[:3], qry[:4], 5)                                                # k clipped to the cloud size
    assert few[0].shape == (4, 5)


def test_sym_eigen3_matches_numpy(O):
    rng = np.random.default_rng(71)
    for it in range(200):
        P = rng.normal(size=(5, 3)) * (10.0 ** rng.uniform(-2, 1))
        if it % 4 == 0:
            P = np.outer(rng.normal(size=5), rng.normal(size=3)) + 1e-3 * rng.normal(size=(5, 3))   # nearly a line, like a corner fit
        c = P - P.mean(0)
        A = c.T @ c / 5.0
        vals, vecs = O.sym_eigen3(A)
        w, V = np.linalg.eigh(A)
        assert np.allclose(vals, w, rtol=1e-9, atol=1e-12 * max(1.0, w[-1]))
        assert np.all(np.diff(vals) >= 0)
        assert np.allclose(vecs.T @ vecs, np.eye(3), atol=1e-9)                            # orthonormal columns
        assert np.allclose(A @ vecs, vecs * vals, atol=1e-9 * max(1.0, w[-1]))
        if w[2] - w[1] > 1e-6 * w[2]:
            assert abs(abs(vecs[:, 2] @ V[:, 2]) - 1.0) < 1e-8                            # principal direction (sign free)


def test_lstsq_5x3_matches_numpy(O):
    rng = np.random.default_rng(72)
    for _ in range(200):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        pts = rng.normal(size=(5, 3)) * 3
        pts -= np.outer(pts @ n - 2.0, n) * 0.98                                          # five points near the plane n . x = 2
        b = -np.ones(5)
        x = O.lstsq_5x3(pts, b)
        ref = np.linalg.lstsq(pts, b, rcond=None)[0]
        assert np.allclose(x, ref, rtol=1e-8, atol=1e-10), (x, ref)


def test_device_atan_restatement_matches_glibc_bit_for_bit():
    """a-loam_amd/csrc/aloam_atan.hpp (branch-free FDLIBM atanf / atan2f, what k_classify evaluates at reference
    src/scanRegistration.cpp:141-142,208) compiled for the host with -ffp-contract=off, against this box's glibc: ~13 million
    values incl. every boundary of the argument reduction and the special cases."""
    import subprocess
    host = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
    r = subprocess.run(["make", "-C", host, "build/test_atan_port"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(host, "build", "test_atan_port")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:]


def test_shared_trig_restatement_is_within_one_ulp_of_glibc():
    """a-loam_amd/csrc/aloam_trig.hpp (FDLIBM acos / sin / cos as plain IEEE operations; what BOTH the device and this oracle
    evaluate inside Eigen's slerp for DISTORTION 1, reference src/laserOdometry.cpp:120, src/lidarFactor.hpp:29,81) against this
    box's glibc, which the reference build under oracle/_ref uses: never more than 1 ulp apart over 29 million arguments."""
    import subprocess
    host = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
    r = subprocess.run(["make", "-C", host, "build/test_trig_port"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(host, "build", "test_trig_port")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:]
    assert float(r.stdout.split()[-1]) > 0.9                                # and identical to glibc in more than 90 % of them


def test_lm_scenarios_reach_every_solver_branch(O, sequence):
    """tests/lm_scenarios.py must drive the oracle's ceres::Solve restatement (oracle_solver.cpp; reference call
    src/laserOdometry.cpp:494-499) through rejected steps and every termination the configuration can reach — the GPU test
    test_lm_branch_coverage compares the device loop with the oracle on exactly these problems."""
    import lm_scenarios
    model, scs = lm_scenarios.build(O, sequence)
    cover = {}
    for sc in scs:
        orc = O.Oracle(n_scans=64, min_range=model.min_range, lm_max_iterations=sc[6], outer_iterations=sc[7])
        st, pose = lm_scenarios.run(orc, sc)
        for b in lm_scenarios.branches(st):
            cover.setdefault(b, []).append(sc[0])
        if "degenerate" in sc[0]:      # non-finite residual: FAILURE before the first iteration, warm start untouched
            q = np.array(sc[4]) / np.linalg.norm(sc[4])
            assert st["termination"] == [5, 5] and st["lm_iterations"] == [0, 0] and np.allclose(pose["q_lc"], q) and np.allclose(pose["t_lc"], sc[5])
    assert {"termination0", "termination1", "termination2", "termination3", "termination5", "rejected_or_invalid"} <= set(cover), cover.keys()
    assert len(cover["rejected_or_invalid"]) >= 3


def test_std_sort_order_restatement_matches_libstdcxx():
    """a-loam_amd/csrc/aloam_stdsort.hpp (the sequence of swaps of libstdc++'s introsort, which decides the order pcl::VoxelGrid sums the members of a
    voxel in; what the reference-order validation mode of the HIP path replays) compiled for the host against this toolchain's std::sort and
    std::partial_sort: the same permutation on 30 000 arrays shaped like VoxelGrid's index vector (few distinct keys, runs, sorted / reversed stretches)."""
    import subprocess
    host = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
    r = subprocess.run(["make", "-C", host, "build/test_stdsort_port"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(host, "build", "test_stdsort_port")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("aloam_stdsort == std::sort"), r.stdout[-2000:]
