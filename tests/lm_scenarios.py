"""Odometry problems built to drive the ceres::Solve stand-in (reference call: src/laserOdometry.cpp:494-499) through the branches
ordinary sweeps never reach: rejected steps (relative decrease <= 1e-3), parameter / function / gradient-tolerance exits, invalid
steps and the FAILURE exit.  Used by the CPU test that pins WHICH branches the oracle takes on them and by the GPU test that
demands the HIP path takes the same ones (tests/test_oracle.py, tests/test_gpu_parity.py).

Every scenario is (features of the current sweep, last corner cloud, last surf cloud, warm start para_q, para_t, lm_max_iterations,
outer_iterations), all derived from one seeded synthetic HDL-64 sequence:
  * warm starts 2 m / 10 degrees off,
  * ground-plane-only planar features (x, y, yaw unobservable) with zero to three corner features,
  * fewer residual rows than the six parameters (rank-deficient J^T J: the damped system is all that keeps the step finite),
  * a last corner cloud whose adjacent rings repeat the same points, so LidarEdgeFactor's |a - b| is 0 and the residual is NaN.
"""
import numpy as np

GOOD = ((0.0, 0.0, 0.01, 1.0), (0.95, 0.02, 0.0))
BAD = ((0.02, 0.03, 0.087, 0.99), (2.0, 1.0, 0.3))            # ~10 degrees, 2.3 m
FAR = ((0.0, 0.0, 0.3, 0.95), (4.0, -3.0, 1.0))


def _ring_sorted(c):
    return c[np.argsort(c[:, 3].astype(int), kind="stable")]


def build(O, sequence):
    scans, R, t, model = sequence("HDL-64", 3, seed=9, columns=512)
    orc = O.Oracle(n_scans=64, min_range=model.min_range)
    feats = []
    for x in scans:
        feats.append(orc.scan_register(x))
        orc.odometry_step()
    f2, corner, surf = feats[2], feats[1]["less_sharp"], feats[1]["less_flat"]

    def subset(n_sharp, n_flat, ground_only, seed):
        rng = np.random.default_rng(seed)
        fl, sh = f2["flat"], f2["sharp"]
        if ground_only:
            fl = fl[fl[:, 2] < -1.2]
        fl = _ring_sorted(fl[rng.permutation(len(fl))[:n_flat]])
        sh = _ring_sorted(sh[rng.permutation(len(sh))[:n_sharp]])
        return {"sharp": sh, "less_sharp": f2["less_sharp"], "flat": fl, "less_flat": f2["less_flat"]}

    out = []
    for lm in (4, 8):
        for name, (q, tt) in (("good", GOOD), ("bad", BAD), ("far", FAR)):
            out.append((f"full-{name}-lm{lm}", f2, corner, surf, q, tt, lm, 2))
        out.append((f"full-good-lm{lm}-outer6", f2, corner, surf, GOOD[0], GOOD[1], lm, 6))
        for seed in range(6):
            for ns, nf, g in ((0, 12, True), (1, 12, True), (2, 6, True), (3, 3, False), (0, 4, True), (1, 2, True)):
                for name, (q, tt) in (("good", GOOD), ("bad", BAD)):
                    out.append((f"sub-s{seed}-c{ns}-p{nf}-{'ground' if g else 'any'}-{name}-lm{lm}", subset(ns, nf, g, seed), corner, surf, q, tt, lm, 2))
    # every last corner point repeated in the next ring: closest point and its adjacent-ring neighbour coincide -> |a - b| = 0
    part = corner[:3600]                                            # twice that stays inside the 120-per-ring capacity of a corner cloud
    dup = part.copy()
    dup[:, 3] += 1.0
    both = _ring_sorted(np.concatenate([part, dup]))
    both = both[both[:, 3] < 63.9]
    for lm in (4, 8):
        out.append((f"degenerate-edges-lm{lm}", f2, both, surf, GOOD[0], GOOD[1], lm, 2))
    return model, out


def run(dev, sc):
    name, f, corner, surf, q, t, lm, outer = sc
    q = np.array(q, float)
    q /= np.linalg.norm(q)
    dev.set_features(f)
    dev.set_last(corner, surf)
    dev.set_state(q, np.array(t, float), [0, 0, 0, 1.0], [0, 0, 0.0], inited=True)
    dev.odometry_step()
    return dev.odom_stats(), dev.pose()


def branches(st):
    """Which solver branches a pair of solves provably went through, from the summary the ABI exposes."""
    seen = set()
    for k in range(2):
        it, ok, term = st["lm_iterations"][k], st["lm_successful"][k], st["termination"][k]
        seen.add(f"termination{term}")
        # a tolerance exit (1, 2) consumes one iteration without a successful step; anything beyond that was rejected or invalid
        spare = it - ok - (1 if term in (1, 2) else 0)
        if spare > 0:
            seen.add("rejected_or_invalid")
        if ok == 0 and it > 0 and not np.isfinite(st["initial_cost"][k]):
            seen.add("invalid")
    return seen
