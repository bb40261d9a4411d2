"""ctypes binding of libaloam_mi355x.so (include/aloam_mi355x.h) for tests and bench.

This is NOT a second implementation: every method is one C-ABI call.  The library itself refuses to work
without a HIP device (ALOAM_E_HIP); loading it and listing its symbols is possible on a CPU-only box.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "lib", "libaloam_mi355x.so")
HEADER_PATH = os.path.join(_ROOT, "include", "aloam_mi355x.h")

CLOUD_FULL, CLOUD_SHARP, CLOUD_LESS_SHARP, CLOUD_FLAT, CLOUD_LESS_FLAT, CLOUD_CORNER_LAST, CLOUD_SURF_LAST = range(7)
E_ARG, E_SCAN_LINES, E_EMPTY, E_CAPACITY, E_HIP, E_STATE = -1, -2, -3, -4, -5, -6
MAP_REGISTERED, MAP_CORNER_STACK, MAP_SURF_STACK = 2, 3, 4
STAGE_REGISTRATION, STAGE_ODOMETRY, STAGE_MAPPING, STAGE_ALL = 1, 2, 4, 7
MAP_INFO_KEYS = ("cenW", "cenH", "cenD", "frame_count", "from_map_corner", "from_map_surf", "corner_stack", "surf_stack",
                 "corner_num0", "corner_num1", "surf_num0", "surf_num1", "lm_iterations0", "lm_iterations1", "termination0", "compactions")


class AloamConfig(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("min_range", C.c_float), ("ring_from_field", C.c_int), ("batch", C.c_int),
                ("max_points", C.c_int), ("max_ring_points", C.c_int), ("device", C.c_int), ("lm_max_iterations", C.c_int),
                ("outer_iterations", C.c_int), ("distortion", C.c_int)]


class AloamOdomStats(C.Structure):
    _fields_ = [("corner_corr", C.c_int * 2), ("plane_corr", C.c_int * 2), ("lm_iterations", C.c_int * 2),
                ("lm_successful", C.c_int * 2), ("initial_cost", C.c_double * 2), ("final_cost", C.c_double * 2),
                ("termination", C.c_int * 2)]


class AloamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"aloam error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with csrc/Makefile (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + [HEADER_PATH]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-j4", "-C", src_dir], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc build failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Every function include/aloam_mi355x.h declares."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aloam_[a-z_0-9]+)\s*\(", txt)))


_lib = None


def lib():
    """Load the shared library; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first; there is no CPU fallback")
        # PyTorch-ROCm wheels bundle their own libamdhip64; if this process loads the system HIP runtime first (through
        # libaloam_mi355x.so) and torch afterwards, torch's copy finds no device.  Loading torch first makes both use one runtime.
        # The C library itself has no torch dependency; this only concerns Python processes that use both.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(os.environ.get("ALOAM_MI355X_LIB", LIB_PATH))   # override: A/B runs of two builds of the same library
        vp, ip = C.c_void_p, C.POINTER(C.c_int)
        L.aloam_default_config.argtypes = [C.POINTER(AloamConfig)]; L.aloam_default_config.restype = None
        L.aloam_create.argtypes = [C.POINTER(AloamConfig), C.POINTER(vp)]
        L.aloam_create_stages.argtypes = [C.POINTER(AloamConfig), C.c_int, C.POINTER(vp)]
        L.aloam_destroy.argtypes = [vp]; L.aloam_destroy.restype = None
        L.aloam_last_error.argtypes = [vp]; L.aloam_last_error.restype = C.c_char_p
        L.aloam_stream.argtypes = [vp]; L.aloam_stream.restype = vp
        L.aloam_synchronize.argtypes = [vp]
        L.aloam_set_voxel_sum_order.argtypes = [vp, C.c_int]
        L.aloam_scan_register.argtypes = [vp, C.POINTER(vp), ip, C.c_int]
        L.aloam_scan_register_device.argtypes = [vp, vp, C.c_longlong, ip, C.c_int]
        L.aloam_scan_register_host.argtypes = [vp, vp, C.c_longlong, ip, C.c_int]
        L.aloam_process_host.argtypes = [vp, vp, C.c_longlong, ip, C.c_int]
        L.aloam_input_consumed.argtypes = [vp]
        L.aloam_odometry_step.argtypes = [vp]
        L.aloam_process_device.argtypes = [vp, vp, C.c_longlong, ip, C.c_int]
        L.aloam_cloud_size.argtypes = [vp, C.c_int, C.c_int]
        L.aloam_get_cloud.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.aloam_get_pose.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.aloam_get_odom_stats.argtypes = [vp, C.c_int, C.POINTER(AloamOdomStats)]
        L.aloam_set_features.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.aloam_set_last.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.aloam_set_state.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.aloam_set_system_inited.argtypes = [vp, C.c_int]
        L.aloam_get_ring_ranges.argtypes = [vp, C.c_int, vp, vp]
        L.aloam_get_curvature.argtypes = [vp, C.c_int, vp, C.c_int]
        L.aloam_get_labels.argtypes = [vp, C.c_int, vp, C.c_int]
        L.aloam_get_last_cloud_order.argtypes = [vp, C.c_int, vp]
        L.aloam_get_correspondences.argtypes = [vp, C.c_int, vp, C.c_int, ip, vp, vp, C.c_int, ip, vp]
        L.aloam_mapping_enable.argtypes = [vp, C.c_float, C.c_float, C.c_int]
        L.aloam_mapping_step.argtypes = [vp]
        L.aloam_mapping_set_pool_limit.argtypes = [vp, C.c_int]
        L.aloam_get_map_pool_info.argtypes = [vp, vp]
        L.aloam_set_map.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]
        L.aloam_set_map_frame.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
        L.aloam_set_full_cloud.argtypes = [vp, C.c_int, vp, C.c_int]
        L.aloam_get_map_pose.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.aloam_get_map_info.argtypes = [vp, C.c_int, vp]
        L.aloam_map_cube_counts.argtypes = [vp, C.c_int, C.c_int, vp]
        L.aloam_get_map_cube.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.aloam_get_map_cloud.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.aloam_profile_enable.argtypes = [vp, C.c_int]
        L.aloam_profile_kernel_count.argtypes = []
        L.aloam_profile_kernel_name.argtypes = [C.c_int]; L.aloam_profile_kernel_name.restype = C.c_char_p
        L.aloam_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Aloam:
    """`batch` independent sequences on one MI355X.  Method names mirror the oracle binding so parity tests read
    `gpu.scan_register(x)` next to `oracle.scan_register(x)`."""

    def __init__(self, n_scans=64, min_range=5.0, ring_from_field=False, batch=1, max_points=140000, max_ring_points=4107,
                 device=0, lm_max_iterations=4, outer_iterations=2, distortion=False, stages=STAGE_ALL):
        L = lib()
        cfg = AloamConfig()
        L.aloam_default_config(C.byref(cfg))
        cfg.n_scans, cfg.min_range, cfg.ring_from_field, cfg.batch = n_scans, min_range, int(ring_from_field), batch
        cfg.max_points, cfg.max_ring_points, cfg.device = max_points, max_ring_points, device
        cfg.lm_max_iterations, cfg.outer_iterations = lm_max_iterations, outer_iterations
        cfg.distortion = int(distortion)
        self.cfg, self.batch, self.n_scans, self.max_points = cfg, batch, n_scans, max_points
        h = C.c_void_p()
        rc = L.aloam_create_stages(C.byref(cfg), int(stages), C.byref(h))
        self.h = h
        if rc != 0:
            msg = L.aloam_last_error(h).decode() if h else "allocation failed"
            if h:
                L.aloam_destroy(h)
            self.h = None
            raise AloamError(rc, msg)

    def set_voxel_sum_order(self, reference_order=True):
        """True: pcl::VoxelGrid's own summation order (libstdc++ std::sort replayed; validation mode); False: input order (default)."""
        self._check(lib().aloam_set_voxel_sum_order(self.h, 1 if reference_order else 0))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.aloam_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise AloamError(rc, lib().aloam_last_error(self.h).decode())
        return rc

    # ---- stage 1 -------------------------------------------------------------------------------------------
    def scan_register(self, scans, check=True):
        """scans: one float32 [N,>=3] array per sequence (or a single array when batch == 1)."""
        if isinstance(scans, np.ndarray) and scans.ndim == 2:
            scans = [scans]
        assert len(scans) == self.batch
        arrs = [_f32(s) for s in scans]
        stride = arrs[0].strides[0] if arrs[0].shape[0] else 4 * arrs[0].shape[1]
        assert all(a.ndim == 2 and a.shape[1] >= 3 and (a.shape[0] == 0 or a.strides[0] == stride) for a in arrs)   # 3 columns = the 12-byte x y z wire format
        ptrs = (C.c_void_p * self.batch)(*[a.ctypes.data for a in arrs])
        nin = (C.c_int * self.batch)(*[a.shape[0] for a in arrs])
        self._check(lib().aloam_scan_register(self.h, ptrs, nin, stride))
        if check:
            self.synchronize()

    def scan_register_device(self, d_ptr, seq_stride_bytes, n_in, stride_bytes=16):
        nin = (C.c_int * self.batch)(*[int(v) for v in n_in])
        self._check(lib().aloam_scan_register_device(self.h, C.c_void_p(d_ptr), seq_stride_bytes, nin, stride_bytes))

    def process_device(self, d_ptr, seq_stride_bytes, n_in, stride_bytes=16):
        nin = n_in if isinstance(n_in, C.Array) else (C.c_int * self.batch)(*[int(v) for v in n_in])
        self._check(lib().aloam_process_device(self.h, C.c_void_p(d_ptr), seq_stride_bytes, nin, stride_bytes))

    def process_host(self, h_ptr, seq_stride_bytes, n_in, stride_bytes=16):
        """One host-resident batch (pinned for true asynchrony): batched H2D copy on the copy stream + stage 1 + stage 2."""
        nin = n_in if isinstance(n_in, C.Array) else (C.c_int * self.batch)(*[int(v) for v in n_in])
        self._check(lib().aloam_process_host(self.h, C.c_void_p(h_ptr), seq_stride_bytes, nin, stride_bytes))

    def scan_register_host(self, h_ptr, seq_stride_bytes, n_in, stride_bytes=16):
        nin = n_in if isinstance(n_in, C.Array) else (C.c_int * self.batch)(*[int(v) for v in n_in])
        self._check(lib().aloam_scan_register_host(self.h, C.c_void_p(h_ptr), seq_stride_bytes, nin, stride_bytes))

    def input_consumed(self):
        self._check(lib().aloam_input_consumed(self.h))

    def synchronize(self):
        self._check(lib().aloam_synchronize(self.h))

    def cloud(self, which, seq=0):
        n = self._check(lib().aloam_cloud_size(self.h, seq, which))
        out = np.zeros((n, 4), np.float32)
        self._check(lib().aloam_get_cloud(self.h, seq, which, _p(out), n))
        return out

    def features(self, seq=0):
        return {"cloud": self.cloud(CLOUD_FULL, seq), "sharp": self.cloud(CLOUD_SHARP, seq), "less_sharp": self.cloud(CLOUD_LESS_SHARP, seq),
                "flat": self.cloud(CLOUD_FLAT, seq), "less_flat": self.cloud(CLOUD_LESS_FLAT, seq)}

    def ring_ranges(self, seq=0):
        s = np.zeros(self.n_scans, np.int32); c = np.zeros(self.n_scans, np.int32)
        self._check(lib().aloam_get_ring_ranges(self.h, seq, _p(s), _p(c)))
        return s, c

    def per_point(self, seq=0):
        n = self._check(lib().aloam_cloud_size(self.h, seq, CLOUD_FULL))
        curv = np.zeros(n, np.float32); lab = np.zeros(n, np.int32)
        self._check(lib().aloam_get_curvature(self.h, seq, _p(curv), n))
        self._check(lib().aloam_get_labels(self.h, seq, _p(lab), n))
        return curv, lab

    # ---- stage 2 -------------------------------------------------------------------------------------------
    def set_features(self, f, seq=0):
        a = [_f32(f[k]) for k in ("sharp", "less_sharp", "flat", "less_flat")]
        self._check(lib().aloam_set_features(self.h, seq, _p(a[0]), len(a[0]), _p(a[1]), len(a[1]), _p(a[2]), len(a[2]), _p(a[3]), len(a[3])))

    def set_last(self, corner_last, surf_last, seq=0):
        a, b = _f32(corner_last), _f32(surf_last)
        self._check(lib().aloam_set_last(self.h, seq, _p(a), len(a), _p(b), len(b)))

    def set_state(self, para_q, para_t, q_w=(0, 0, 0, 1), t_w=(0, 0, 0), seq=0, inited=True):
        a, b, c, d = _f64(para_q), _f64(para_t), _f64(q_w), _f64(t_w)
        self._check(lib().aloam_set_state(self.h, seq, _p(a), _p(b), _p(c), _p(d)))
        self._check(lib().aloam_set_system_inited(self.h, int(inited)))

    def odometry_step(self):
        self._check(lib().aloam_odometry_step(self.h))

    def pose(self, seq=0):
        qw, tw, ql, tl = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        self._check(lib().aloam_get_pose(self.h, seq, _p(qw), _p(tw), _p(ql), _p(tl)))
        return {"q_w": qw, "t_w": tw, "q_lc": ql, "t_lc": tl}

    def odom_stats(self, seq=0):
        st = AloamOdomStats()
        self._check(lib().aloam_get_odom_stats(self.h, seq, C.byref(st)))
        return {k: list(getattr(st, k)) for k, _ in AloamOdomStats._fields_}

    def last_cloud_order(self, seq=0):
        v = np.zeros(2, np.int32)
        self._check(lib().aloam_get_last_cloud_order(self.h, seq, _p(v)))
        return int(v[0]), int(v[1])

    def correspondences(self, seq=0):
        cap_e, cap_p = self.n_scans * 12, self.n_scans * 24
        e = np.zeros((cap_e, 9), np.float32); p = np.zeros((cap_p, 12), np.float32)
        eq = np.zeros(cap_e, np.int32); pq = np.zeros(cap_p, np.int32)
        ne, npl = C.c_int(0), C.c_int(0)
        self._check(lib().aloam_get_correspondences(self.h, seq, _p(e), cap_e, C.byref(ne), _p(eq), _p(p), cap_p, C.byref(npl), _p(pq)))
        return e[:ne.value].copy(), p[:npl.value].copy(), eq[:ne.value].copy(), pq[:npl.value].copy()

    # ---- stage 3 -------------------------------------------------------------------------------------------
    def mapping_enable(self, line_res=0.4, plane_res=0.8, pool_points=262144, pool_limit=None):
        """pool_points: where the map pools start (they double as the map grows); pool_limit: the ceiling (None = the library's default)."""
        if pool_limit is not None:
            self._check(lib().aloam_mapping_set_pool_limit(self.h, int(pool_limit)))
        self._check(lib().aloam_mapping_enable(self.h, float(line_res), float(plane_res), int(pool_points)))

    def map_pool_info(self):
        v = np.zeros(4, np.int32)
        self._check(lib().aloam_get_map_pool_info(self.h, _p(v)))
        return {"pool_points": int(v[0]), "growths": int(v[1]), "limit": int(v[2]), "live_max": int(v[3])}

    def set_map(self, cubes, cls, seq=0):
        """cubes: {cube index: (n, 4) points} -> laserCloudCornerArray (cls 0) / laserCloudSurfArray (cls 1) of the sequence."""
        ids = np.array(sorted(cubes), np.int32)
        cnt = np.array([len(cubes[int(i)]) for i in ids], np.int32)
        pts = _f32(np.concatenate([cubes[int(i)] for i in ids])) if len(ids) else np.zeros((0, 4), np.float32)
        self._check(lib().aloam_set_map(self.h, seq, cls, _p(ids), _p(cnt), len(ids), _p(pts)))

    def set_map_frame(self, cen, q_wmap_wodom, t_wmap_wodom, frame_count, seq=0):
        a, q, t = np.ascontiguousarray(cen, dtype=np.int32), _f64(q_wmap_wodom), _f64(t_wmap_wodom)
        self._check(lib().aloam_set_map_frame(self.h, seq, _p(a), _p(q), _p(t), int(frame_count)))

    def mapping_step(self):
        self._check(lib().aloam_mapping_step(self.h))

    def set_full_cloud(self, cloud, seq=0):
        a = _f32(cloud)
        self._check(lib().aloam_set_full_cloud(self.h, seq, _p(a), len(a)))

    def mapping_step_inputs(self, q_wodom, t_wodom, corner_last, surf_last, full_res, seq=0):
        """Teacher-forced frame: inject what the mapping node receives (reference src/laserMapping.cpp:175-228), then step."""
        self.set_last(corner_last, surf_last, seq)
        self.set_full_cloud(full_res, seq)
        p = self.pose(seq)
        self.set_state(p["q_lc"], p["t_lc"], q_wodom, t_wodom, seq)
        self.mapping_step()
        return self.map_pose(seq)

    def map_pose(self, seq=0):
        qw, tw, qm, tm = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        self._check(lib().aloam_get_map_pose(self.h, seq, _p(qw), _p(tw), _p(qm), _p(tm)))
        return {"q_w": qw, "t_w": tw, "q_wmap_wodom": qm, "t_wmap_wodom": tm}

    def map_info(self, seq=0):
        v = np.zeros(16, np.int32)
        self._check(lib().aloam_get_map_info(self.h, seq, _p(v)))
        return dict(zip(MAP_INFO_KEYS, (int(x) for x in v)))

    def map_cubes(self, cls, seq=0):
        cnt = np.zeros(21 * 21 * 11, np.int32)
        self._check(lib().aloam_map_cube_counts(self.h, seq, cls, _p(cnt)))
        out = {}
        for i in np.nonzero(cnt)[0]:
            pts = np.zeros((int(cnt[i]), 4), np.float32)
            self._check(lib().aloam_get_map_cube(self.h, seq, cls, int(i), _p(pts), len(pts)))
            out[int(i)] = pts
        return out

    def map_cloud(self, which, seq=0):
        n = self._check(lib().aloam_get_map_cloud(self.h, seq, which, None, 0))
        out = np.zeros((n, 4), np.float32)
        if n:
            self._check(lib().aloam_get_map_cloud(self.h, seq, which, _p(out), n))
        return out

    # ---- profiling -----------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(lib().aloam_profile_enable(self.h, int(on)))

    def profile(self):
        out = {}
        for k in range(lib().aloam_profile_kernel_count()):
            ms, n, by = C.c_double(0), C.c_longlong(0), C.c_double(0)
            self._check(lib().aloam_profile_get(self.h, k, C.byref(ms), C.byref(n), C.byref(by)))
            out[lib().aloam_profile_kernel_name(k).decode()] = {"total_ms": ms.value, "launches": n.value, "bytes_per_launch": by.value}
        return out

    def stream(self):
        return lib().aloam_stream(self.h)
