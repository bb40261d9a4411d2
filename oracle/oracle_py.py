"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — nowhere else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")


class OrcConfig(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("min_range", C.c_float), ("ring_from_field", C.c_int), ("canonical_order", C.c_int),
                ("nn_brute", C.c_int), ("analytic_jacobian", C.c_int), ("apply_converged_step", C.c_int),
                ("lm_max_iterations", C.c_int), ("outer_iterations", C.c_int), ("distortion", C.c_int)]


class OrcOdomStats(C.Structure):
    _fields_ = [("corner_corr", C.c_int * 2), ("plane_corr", C.c_int * 2), ("lm_iterations", C.c_int * 2),
                ("lm_successful", C.c_int * 2), ("initial_cost", C.c_double * 2), ("final_cost", C.c_double * 2),
                ("termination", C.c_int * 2)]


CLOUD_FULL, CLOUD_SHARP, CLOUD_LESS_SHARP, CLOUD_FLAT, CLOUD_LESS_FLAT, CLOUD_CORNER_LAST, CLOUD_SURF_LAST = range(7)


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (g++ only)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp", ".h"))]
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, ip, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.orc_default_config.argtypes = [C.POINTER(OrcConfig)]
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]; L.orc_create.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_last_error.argtypes = [vp]; L.orc_last_error.restype = C.c_char_p
        L.orc_scan_register.argtypes = [vp, vp, C.c_int, C.c_int]
        L.orc_cloud_size.argtypes = [vp, C.c_int]
        L.orc_get_cloud.argtypes = [vp, C.c_int, vp, C.c_int]
        L.orc_get_ring_ranges.argtypes = [vp, vp, vp]
        L.orc_get_curvature.argtypes = [vp, vp, C.c_int]
        L.orc_get_labels.argtypes = [vp, vp, C.c_int]
        L.orc_get_picked.argtypes = [vp, vp, C.c_int]
        L.orc_set_features.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.orc_set_last.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.orc_set_state.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.orc_odometry_step.argtypes = [vp]
        L.orc_get_pose.argtypes = [vp, vp, vp, vp, vp]
        L.orc_get_odom_stats.argtypes = [vp, C.POINTER(OrcOdomStats)]
        L.orc_get_correspondences.argtypes = [vp, vp, C.c_int, ip, vp, C.c_int, ip, vp, vp]
        L.orc_voxel_filter.argtypes = [vp, C.c_int, C.c_float, C.c_int, vp, C.c_int]
        L.orc_nn_search.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.orc_factor_eval.argtypes = [C.c_int, vp, vp, vp, C.c_int, vp, vp]
        L.orc_factor_eval_s.argtypes = [C.c_int, vp, C.c_double, vp, vp, vp, vp]
        L.orc_cost.argtypes = [C.c_int, vp, C.c_int, vp, vp, vp]; L.orc_cost.restype = C.c_double
        L.orc_lm_solve.argtypes = [C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, ip]
        L.orc_atan2f_port.argtypes = [C.c_float, C.c_float]; L.orc_atan2f_port.restype = C.c_float
        L.orc_quat_plus.argtypes = [vp, vp, vp]
        L.orc_map_config.argtypes = [vp, C.c_float, C.c_float]
        L.orc_mapping_step.argtypes = [vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.orc_map_get_pose.argtypes = [vp, vp, vp, vp, vp]
        L.orc_map_get_info.argtypes = [vp, vp]
        L.orc_map_cloud_size.argtypes = [vp, C.c_int, C.c_int]
        L.orc_map_get_cloud.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.orc_map_cube_counts.argtypes = [vp, C.c_int, vp]
        L.orc_knn_search.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_sym_eigen3.argtypes = [vp, vp, vp]; L.orc_sym_eigen3.restype = None
        L.orc_lstsq_5x3.argtypes = [vp, vp, vp]; L.orc_lstsq_5x3.restype = None
        L.orc_decision_log.argtypes = [C.c_int]; L.orc_decision_log.restype = None
        L.orc_decision_log_size.argtypes = []; L.orc_decision_log_size.restype = C.c_longlong
        L.orc_decision_log_fetch.argtypes = [vp, vp, vp, C.c_longlong]; L.orc_decision_log_fetch.restype = C.c_longlong
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One sequence: scan registration + odometry state, mirroring the two reference nodes."""

    def __init__(self, n_scans=64, min_range=5.0, ring_from_field=False, canonical_order=True, nn_brute=False,
                 analytic_jacobian=False, apply_converged_step=False, lm_max_iterations=4, outer_iterations=2, distortion=False):
        L = lib()
        cfg = OrcConfig()
        L.orc_default_config(C.byref(cfg))
        cfg.n_scans, cfg.min_range, cfg.ring_from_field = n_scans, min_range, int(ring_from_field)
        cfg.canonical_order, cfg.nn_brute, cfg.analytic_jacobian = int(canonical_order), int(nn_brute), int(analytic_jacobian)
        cfg.apply_converged_step, cfg.lm_max_iterations, cfg.outer_iterations = int(apply_converged_step), lm_max_iterations, outer_iterations
        cfg.distortion = int(distortion)
        self.cfg = cfg
        self.n_scans = n_scans
        self.h = L.orc_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            try:
                _lib.orc_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def scan_register(self, pts):
        pts = _f32(pts)
        assert pts.ndim == 2 and pts.shape[1] >= 4
        rc = lib().orc_scan_register(self.h, _p(pts), pts.shape[0], pts.strides[0])
        if rc != 0:
            raise RuntimeError(lib().orc_last_error(self.h).decode())
        return self.features()

    def cloud(self, which):
        n = lib().orc_cloud_size(self.h, which)
        out = np.zeros((max(n, 0), 4), np.float32)
        lib().orc_get_cloud(self.h, which, _p(out), n)
        return out

    def features(self):
        return {"cloud": self.cloud(CLOUD_FULL), "sharp": self.cloud(CLOUD_SHARP), "less_sharp": self.cloud(CLOUD_LESS_SHARP),
                "flat": self.cloud(CLOUD_FLAT), "less_flat": self.cloud(CLOUD_LESS_FLAT)}

    def ring_ranges(self):
        s = np.zeros(self.n_scans, np.int32); c = np.zeros(self.n_scans, np.int32)
        lib().orc_get_ring_ranges(self.h, _p(s), _p(c))
        return s, c

    def per_point(self):
        n = lib().orc_cloud_size(self.h, CLOUD_FULL)
        curv = np.zeros(n, np.float32); lab = np.zeros(n, np.int32); pk = np.zeros(n, np.int32)
        lib().orc_get_curvature(self.h, _p(curv), n)
        lib().orc_get_labels(self.h, _p(lab), n)
        lib().orc_get_picked(self.h, _p(pk), n)
        return curv, lab, pk

    def set_features(self, f):
        a = [_f32(f[k]) for k in ("sharp", "less_sharp", "flat", "less_flat")]
        lib().orc_set_features(self.h, _p(a[0]), len(a[0]), _p(a[1]), len(a[1]), _p(a[2]), len(a[2]), _p(a[3]), len(a[3]))

    def set_last(self, corner_last, surf_last):
        a, b = _f32(corner_last), _f32(surf_last)
        lib().orc_set_last(self.h, _p(a), len(a), _p(b), len(b))

    def set_state(self, para_q, para_t, q_w=(0, 0, 0, 1), t_w=(0, 0, 0), inited=True):
        a, b, c, d = _f64(para_q), _f64(para_t), _f64(q_w), _f64(t_w)
        lib().orc_set_state(self.h, _p(a), _p(b), _p(c), _p(d), int(inited))

    def odometry_step(self):
        rc = lib().orc_odometry_step(self.h)
        if rc != 0:
            raise RuntimeError(lib().orc_last_error(self.h).decode())
        return self.pose()

    def pose(self):
        qw, tw, ql, tl = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        lib().orc_get_pose(self.h, _p(qw), _p(tw), _p(ql), _p(tl))
        return {"q_w": qw, "t_w": tw, "q_lc": ql, "t_lc": tl}

    def odom_stats(self):
        st = OrcOdomStats()
        lib().orc_get_odom_stats(self.h, C.byref(st))
        return {k: list(getattr(st, k)) for k, _ in OrcOdomStats._fields_}

    def correspondences(self):
        cap_e, cap_p = 16384, 32768
        e = np.zeros((cap_e, 9)); p = np.zeros((cap_p, 12))
        ne, npl = C.c_int(0), C.c_int(0)
        eq = np.zeros(cap_e, np.int32); pq = np.zeros(cap_p, np.int32)
        lib().orc_get_correspondences(self.h, _p(e), cap_e, C.byref(ne), _p(p), cap_p, C.byref(npl), _p(eq), _p(pq))
        return e[:ne.value].copy(), p[:npl.value].copy(), eq[:ne.value].copy(), pq[:npl.value].copy()


MAP_CORNER_CUBE, MAP_SURF_CUBE, MAP_REGISTERED, MAP_CORNER_STACK, MAP_SURF_STACK = range(5)
MAP_INFO_KEYS = ("cenW", "cenH", "cenD", "frame_count", "from_map_corner", "from_map_surf", "corner_stack", "surf_stack",
                 "corner_num0", "corner_num1", "surf_num0", "surf_num1", "lm_iterations0", "lm_iterations1", "termination0", "termination1")


def _map_methods():
    def map_config(self, line_res, plane_res):
        lib().orc_map_config(self.h, float(line_res), float(plane_res))

    def mapping_step(self, q_wodom, t_wodom, corner_last, surf_last, full_res):
        """One pass of process() (laserMapping.cpp:231-893) on what the node receives for one frame."""
        q, t = _f64(q_wodom), _f64(t_wodom)
        c, s, f = _f32(corner_last), _f32(surf_last), _f32(full_res)
        rc = lib().orc_mapping_step(self.h, _p(q), _p(t), _p(c), len(c), _p(s), len(s), _p(f), len(f))
        if rc:
            raise RuntimeError(lib().orc_last_error(self.h).decode())
        return self.map_pose()

    def map_pose(self):
        qw, tw, qm, tm = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        lib().orc_map_get_pose(self.h, _p(qw), _p(tw), _p(qm), _p(tm))
        return {"q_w": qw, "t_w": tw, "q_wmap_wodom": qm, "t_wmap_wodom": tm}

    def map_info(self):
        v = np.zeros(16, np.int32)
        lib().orc_map_get_info(self.h, _p(v))
        return dict(zip(MAP_INFO_KEYS, (int(x) for x in v)))

    def map_cloud(self, which, cube=0):
        n = lib().orc_map_cloud_size(self.h, which, cube)
        out = np.zeros((max(n, 0), 4), np.float32)
        if n > 0:
            lib().orc_map_get_cloud(self.h, which, cube, _p(out), n)
        return out

    def map_cubes(self, cls):
        """{cube index: (n, 4) points} of the non-empty cubes of class 0 (corner) / 1 (surf)."""
        cnt = np.zeros(21 * 21 * 11, np.int32)
        lib().orc_map_cube_counts(self.h, cls, _p(cnt))
        return {int(i): self.map_cloud(cls, int(i)) for i in np.nonzero(cnt)[0]}

    return dict(map_config=map_config, mapping_step=mapping_step, map_pose=map_pose, map_info=map_info, map_cloud=map_cloud, map_cubes=map_cubes)


def knn_search(target, query, k=5, brute=False):
    t, q = _f32(target), _f32(query)
    idx = np.zeros((len(q), k), np.int32); d2 = np.zeros((len(q), k), np.float32)
    lib().orc_knn_search(_p(t), len(t), _p(q), len(q), k, int(brute), _p(idx), _p(d2))
    return idx, d2


def sym_eigen3(A):
    A = _f64(A); vals = np.zeros(3); vecs = np.zeros((3, 3))
    lib().orc_sym_eigen3(_p(A), _p(vals), _p(vecs))
    return vals, vecs


def lstsq_5x3(A, b):
    A, b = _f64(A), _f64(b); x = np.zeros(3)
    lib().orc_lstsq_5x3(_p(A), _p(b), _p(x))
    return x


def voxel_filter(xyzi, leaf, canonical=True):
    a = _f32(xyzi)
    out = np.zeros_like(a)
    n = lib().orc_voxel_filter(_p(a), len(a), leaf, int(canonical), _p(out), len(a))
    return out[:n].copy()


def nn_search(target, query, brute=False):
    t, q = _f32(target), _f32(query)
    idx = np.zeros(len(q), np.int32); d2 = np.zeros(len(q), np.float32)
    lib().orc_nn_search(_p(t), len(t), _p(q), len(q), int(brute), _p(idx), _p(d2))
    return idx, d2


def factor_eval(kind, consts, q, t, analytic):
    c, q, t = _f64(consts), _f64(q), _f64(t)
    rows = 3 if kind == 0 else 1
    r = np.zeros(rows); J = np.zeros((rows, 6))
    lib().orc_factor_eval(kind, _p(c), _p(q), _p(t), int(analytic), _p(r), _p(J))
    return r, J


def factor_eval_s(kind, consts, s, q, t):
    """Residual + Jacobian (projected through the quaternion Plus Jacobian) of one factor with interpolation ratio s."""
    c, q, t = _f64(consts), _f64(q), _f64(t)
    rows = 3 if kind == 0 else 1
    r = np.zeros(rows); J = np.zeros((rows, 6))
    lib().orc_factor_eval_s(kind, _p(c), float(s), _p(q), _p(t), _p(r), _p(J))
    return r, J


def cost(edges, planes, q, t):
    e, p, q, t = _f64(edges).reshape(-1, 9), _f64(planes).reshape(-1, 12), _f64(q), _f64(t)
    return lib().orc_cost(len(e), _p(e), len(p), _p(p), _p(q), _p(t))


def lm_solve(edges, planes, q, t, max_iterations=4, analytic=False, apply_converged_step=False):
    e, p = _f64(edges).reshape(-1, 9), _f64(planes).reshape(-1, 12)
    q, t = _f64(q).copy(), _f64(t).copy()
    it, su, term = C.c_int(0), C.c_int(0), C.c_int(0)
    c0, c1 = C.c_double(0), C.c_double(0)
    lib().orc_lm_solve(len(e), _p(e), len(p), _p(p), _p(q), _p(t), max_iterations, int(analytic), int(apply_converged_step),
                       C.byref(it), C.byref(su), C.byref(c0), C.byref(c1), C.byref(term))
    return q, t, {"iterations": it.value, "successful": su.value, "initial_cost": c0.value, "final_cost": c1.value, "termination": term.value}


def atan2f_port(y, x):
    return lib().orc_atan2f_port(float(y), float(x))


def quat_plus(q, delta):
    q, d, out = _f64(q), _f64(delta), np.zeros(4)
    lib().orc_quat_plus(_p(q), _p(d), _p(out))
    return out


for _name, _fn in _map_methods().items():
    setattr(Oracle, _name, _fn)


DECISION_KINDS = ("curvature > 0.1", "curvature < 0.1", "gap^2 > 0.05", "range^2 < min_range^2", "odometry d2 < 25", "map 5th neighbour d2 < 1",
                  "eigenvalue ratio > 3", "plane residual > 0.2", "LM parameter tolerance", "LM function tolerance", "LM step acceptance (rel > 1e-3)",
                  "LM gradient tolerance", "LM model change > 0")


def decision_log(enable=True):
    """Start (and clear) or stop the calling thread's decision log."""
    lib().orc_decision_log(int(enable))


def decisions():
    """(kinds int32[n], values float64[n], thresholds float64[n]) recorded since decision_log(True)."""
    n = lib().orc_decision_log_size()
    k = np.zeros(n, np.int32); v = np.zeros(n); t = np.zeros(n)
    lib().orc_decision_log_fetch(_p(k), _p(v), _p(t), n)
    return k, v, t
