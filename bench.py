#!/usr/bin/env python
"""bench.py — HDL-64 scans/sec of the MI355X A-LOAM hot path (scan registration + scan-to-scan odometry).

One "step" = one sweep of every one of `--batch` independent sequences per GPU pushed through
aloam_process_device(): stage 1 (reference src/scanRegistration.cpp:127-411) + stage 2 (reference
src/laserOdometry.cpp:265-506,554-568).  Headline workload = BASELINE.json configs[1] with synthetic data (KITTI is
not in the image): 64 rings x 2048 columns = 131072 points per sweep, ring-major like KITTI .bin files, minimum_range 5,
noise sigma 0.02 m, sensor driven 1 m per sweep on a 30 m circle.  Inputs are resident in HBM before the timed
region; every sequence replays its `--frames` stored sweeps forwards and backwards (consecutive sweeps are always
1 m apart, so the odometry problem is the real one at every step).

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* exported, rendezvous on 127.0.0.1); under `python -m torch.distributed.run ... bench.py --gpus N` the ranks already
exist and are used as they are.  Sequences are independent, each rank owns `--batch` of them, there is no data-path
collective; a barrier brackets the timed region and the max time over ranks is used ("scaling": "weak").

Prints ONE JSON line on rank 0 (contract in the task statement).  Beside the headline (`value`) it carries, at N = 1:
  roofline         - dominant kernel: algorithmic bytes per launch / its mean hipEvent duration, vs 8 TB/s HBM
  value_host_input - the same steps fed from pinned HOST memory: one batched H2D copy per step on a copy stream, double-buffered
                     against the kernels (aloam_process_host) — the PCIe-inclusive rate, never the headline
  workloads        - BASELINE.json configs[2] (odometry + laserMapping refinement every sweep) and configs[3] (128 x 2048
                     stress), each with its own ms_per_step / roofline
  latency          - single-sensor (batch 1) milliseconds per sweep through the host-buffer entry points the ROS shims use
  cpu_baseline     - the CPU oracle (oracle/, a restatement of the reference; "kind": "port"): one thread (median / p95 per
                     sweep) and one process per host core (sequence-parallel), timed in the same run
  accuracy         - per-sweep pose difference GPU vs oracle (parity) and trajectory ATE vs the synthetic ground truth
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s achievable float4 copy)
# VALU ceiling: 256 CUs x 4 SIMDs, 2.4 GHz.  The integer / select / DPP mix of these kernels occupies one issue slot of four cycles per
# wave-instruction (SQ_ACTIVE_INST_VALU, which counts quad-cycles, equals SQ_INSTS_VALU to 1 % on every kernel here); MI355X_MICROARCH.md
# measures 2 cycles for v_fma_f32 — the rate a pure f32 FMA stream would reach — so both ceilings are printed.
VALU_SIMDS, VALU_CLK_GHZ = 1024, 2.4
VALU_PEAK_GINST = VALU_SIMDS * VALU_CLK_GHZ / 4.0
STAGE_KERNELS = {"map_associate": ["k_map_search<0>", "k_map_search<1>", "k_map_fit<0>", "k_map_fit<1>"],   # a profiled stage = these kernels, once each
                 "map_solve": ["k_map_solve"], "map_register": ["k_map_register"], "map_begin": ["k_map_begin"],
                 "map_grid": ["k_mapgrid_build"], "k_build_grids": ["k_build_grids_fused", "k_build_grids"]}
RK_NAMES = {"k_associate[plane]": "k_associate_pair<true, ", "k_associate[corner]": "k_associate_pair<false, ",      # profiled name -> prefix of the
            "k_ring_features": "k_ring_features<", "k_solve": "k_solve<false>"}                                     # kernel name rocprofv3 reports


def quat_angle(qa, qb):
    """Rotation angle between two xyzw quaternions."""
    d = abs(float(np.dot(qa, qb))) / (np.linalg.norm(qa) * np.linalg.norm(qb))
    return 2.0 * float(np.arccos(min(1.0, d)))


def frame_order(n_frames, n_steps):
    """0,1,..,T-1,T-2,..,1,0,1,..  (ping-pong)"""
    out, t, d = [], 0, 1
    for _ in range(n_steps):
        out.append(t)
        if n_frames > 1:
            if t + d < 0 or t + d >= n_frames:
                d = -d
            t += d
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps of the headline leg (120 x ~9 ms: a timed region above one second); the other legs use min(steps, --extra-steps)")
    ap.add_argument("--extra-steps", type=int, default=30)
    ap.add_argument("--repeat-to-seconds", type=float, default=1.0, help="after the K timed steps, time further identical K-step blocks until this many seconds are covered (0: off; always off with --no-extras); reported as repeated_blocks, never as `value`")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="independent sequences per GPU of the headline leg (round 6, kernel time per sequence: 1024 -> 8.72 us, 2048 -> 8.27, 4096 -> 8.27; 30 MB of HBM per sequence); the configs[2] / [3] comparison legs stay at min(batch, 1024)")
    ap.add_argument("--frames", type=int, default=6, help="stored sweeps per sequence (replayed ping-pong)")
    ap.add_argument("--sensor", default="HDL-64", help="headline workload sensor (HDL-64 = BASELINE configs[1]; ROWS128 = configs[3])")
    ap.add_argument("--mapping", action="store_true", help="headline workload = BASELINE configs[2]: scan-to-map refinement after every sweep")
    ap.add_argument("--map-pool", type=int, default=262144, help="device map capacity per sequence and class the contexts START with (points; the pools double as the maps grow)")
    ap.add_argument("--map-batch", type=int, default=512, help="sequences per GPU of the configs[2] workload (travelling sensor, --map-frames distinct sweeps each)")
    ap.add_argument("--map-frames", type=int, default=100)
    ap.add_argument("--map-warmup", type=int, default=80, help="untimed steps of the configs[2] workload: 125 m of travel, after which the submap is stationary")
    ap.add_argument("--contexts", type=int, default=1, help="split the batch over this many contexts (= HIP streams) on the same GPU")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each CPU baseline leg (one thread; one process per core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip host-input rate, configs[2]/[3] sub-workloads and latency")
    ap.add_argument("--host-input", action="store_true", help="(kept for compatibility: the host-fed rate is part of the default line)")
    ap.add_argument("--latency-sweeps", type=int, default=120)
    ap.add_argument("--travel", action="store_true", help="odometry-only workload on the travelling drive (the configs[2] input: 1.6 m per sweep down a street, 120 m range) instead of laps of the 30 m circle; --frames distinct sweeps per sequence")
    ap.add_argument("--reference-order", action="store_true", help="run the workload of this call in the reference's voxel summation order (aloam_set_voxel_sum_order: the validation mode, std::sort replayed on the device) and say so in config")
    ap.add_argument("--rough", action="store_true", help="KITTI-shaped irregular sweeps (random no-returns, ragged rings, noisy sectors, repeated returns) instead of the clean synthetic ones")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
SHARED_GPU_ENV = "ALOAM_BENCH_SHARED_GPU"   # test hook: let the ranks share the visible devices (control plane on gloo) on a 1-GPU box
FORCE_DIST_ENV = "ALOAM_BENCH_FORCE_DIST"    # test hook: initialise the process group (nccl = RCCL) and run barrier + MAX even when WORLD_SIZE is 1
RANK_ENV_ONLY = "ALOAM_BENCH_RANK_ENV_ONLY"  # test hook: a started rank prints the environment it was given and exits (no GPU needed)


def rank_envs(n, port, base=None):
    """Environment of each of the n ranks `python bench.py --gpus n` starts (what torch.distributed.run would export)."""
    base = dict(os.environ if base is None else base)
    return [dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")) for r in range(n)]


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU)."""
    n = args.gpus
    if not (os.environ.get(SHARED_GPU_ENV) or os.environ.get(RANK_ENV_ONLY)):
        import torch
        have = torch.cuda.device_count()
        assert have >= n, f"--gpus {n} but only {have} HIP device(s) are visible"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for env in rank_envs(n, port):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


# ---------------------------------------------------------------------------------------------------------------------
class Workload:
    """Synthetic sweeps of B sequences x T frames, resident in HBM, and the bookkeeping to replay them."""

    def __init__(self, syn, torch, sensor, B, T, rank, dev, rough=False):
        self.sensor, self.B, self.T, self.rough = sensor, B, T, rough
        self.model = syn.sensor_model(sensor, device=dev)
        self.NP = self.model.dirs.shape[0]
        t0 = time.time()
        self.data = torch.zeros((B, T, self.NP, 4), dtype=torch.float32, device=dev)
        self.counts = np.zeros((B, T), np.int32)
        worlds = [syn.make_world(100 + w).to(dev) for w in range(8)]
        self.gt = {}
        for b in range(B):
            gseq = rank * B + b
            R, tt = syn.trajectory(T, step=1.0, seed=gseq, start_angle=0.37 * gseq)
            gen = torch.Generator(device=dev).manual_seed(9000 + gseq)
            for k in range(T):
                s = syn.render_scan(worlds[gseq % len(worlds)], self.model, R[k], tt[k], 0.02, gen, rough=rough)
                self.counts[b, k] = s.shape[0]
                self.data[b, k, : s.shape[0]] = s
            if b < 4:
                self.gt[b] = (R.numpy(), tt.numpy())
        torch.cuda.synchronize()
        self.gen_s = time.time() - t0
        self.seq_stride = T * self.NP * 16

    def ctx(self, binding, batch, device, **kw):
        m = self.model
        return binding.Aloam(n_scans=m.n_scans, min_range=m.min_range, ring_from_field=m.ring_from_field, batch=batch,
                             max_points=self.NP, max_ring_points=2059 if m.columns <= 2048 else 4107, device=device, **kw)

    def nin(self, k, lo=0, hi=None):
        hi = self.B if hi is None else hi
        return (ctypes.c_int * (hi - lo))(*[int(v) for v in self.counts[lo:hi, k]])

    def describe(self, mapping):
        m = self.model
        return f"synthetic {self.sensor} {m.n_scans}x{m.columns} ({self.NP} pts/sweep{', rough: dropouts / ragged rings / repeated returns' if self.rough else ''}), " + (
            "odometry + laserMapping scan-to-map refinement every sweep" if mapping else "odometry only (scan registration + scan-to-scan odometry, no laserMapping)")


class TravelWorkload(Workload):
    """BASELINE.json configs[2] at steady-state map depth: every sequence DRIVES (a-loam_amd/synthetic.py `travel`: 1.6 m per sweep down a street, never
    returning, 120 m sensor range), T distinct sweeps each, so that after the warm-up the submap the scan-to-map search runs on is the stationary one
    of a moving sensor (everything mapped within the 5 x 5 x 3 cube window, reference src/laserMapping.cpp:509-539) instead of what six sweeps leave
    behind.  `distinct` different drives are rendered and replicated over the batch (identical work per replica, separate memory)."""

    def __init__(self, syn, torch, B, T, rank, dev, distinct=32, step=1.6):
        self.sensor, self.B, self.T, self.rough = "HDL-64", B, T, False
        self.model = syn.sensor_model("HDL-64", device=dev)
        self.NP = self.model.dirs.shape[0]
        t0 = time.time()
        distinct = min(distinct, B)
        assert B % distinct == 0
        self.data = torch.zeros((B, T, self.NP, 4), dtype=torch.float32, device=dev)
        self.counts = np.zeros((B, T), np.int32)
        self.gt = {}
        for b in range(distinct):
            gseq = rank * distinct + b
            world = syn.make_street_world(300 + gseq % 8).to(dev)
            R, tt = syn.trajectory_travel(T, step=step, seed=gseq)
            gen = torch.Generator(device=dev).manual_seed(9500 + gseq)
            for k in range(T):
                s = syn.render_scan(world, self.model, R[k], tt[k], 0.02, gen, max_range=syn.STREET_MAX_RANGE, cull=True)
                self.counts[b, k] = s.shape[0]
                self.data[b, k, : s.shape[0]] = s
            if b < 4:
                self.gt[b] = (R.numpy(), tt.numpy())
        for r in range(1, B // distinct):
            self.data[r * distinct:(r + 1) * distinct] = self.data[:distinct]
            self.counts[r * distinct:(r + 1) * distinct] = self.counts[:distinct]
        torch.cuda.synchronize()
        self.gen_s = time.time() - t0
        self.seq_stride = T * self.NP * 16
        self.distinct, self.step = distinct, step

    def describe(self, mapping):
        m = self.model
        return (f"synthetic {self.sensor} {m.n_scans}x{m.columns} ({self.NP} pts/sweep), sensor TRAVELLING {self.step} m per sweep down a street ({self.T} distinct sweeps per "
                f"sequence, {self.distinct} distinct drives replicated over the batch), " + ("odometry + laserMapping scan-to-map refinement every sweep at steady-state submap depth"
                                                                                             if mapping else "odometry only (scan registration + scan-to-scan odometry, no laserMapping)"))


def map_state(cx, n_seq=4):
    """What the scan-to-map search of the last step ran on (mean over a few sequences) and the state of the pools."""
    infos = [cx.map_info(b) for b in range(min(n_seq, cx.batch))]
    out = {k: int(round(float(np.mean([i[k] for i in infos])))) for k in ("frame_count", "from_map_corner", "from_map_surf", "corner_stack", "surf_stack", "corner_num1", "surf_num1")}
    cnt = np.zeros(21 * 21 * 11, np.int32)
    binding = importlib.import_module("a-loam_amd.binding")
    tot = []
    for cls in (0, 1):
        binding.lib().aloam_map_cube_counts(cx.h, 0, cls, binding._p(cnt))
        tot.append((int(cnt.sum()), int((cnt > 0).sum())))
    out.update({"map_points_corner": tot[0][0], "map_points_surf": tot[1][0], "occupied_cubes_corner": tot[0][1], "occupied_cubes_surf": tot[1][1], "pool": cx.map_pool_info()})
    return out


def timed_resident(torch, dist, world, ctxs, wl, steps, warmup, mapping, repeat_to_s=0.0, repeats_out=None, batch=None):
    """W untimed + exactly K timed steps with the inputs resident in HBM; per-kernel hipEvents on the context's own stream.
    `repeat_to_s` > 0: after the K timed steps (which alone make `value`), further identical blocks of exactly K steps are timed the
    same way (barrier + synchronize on both sides, MAX over ranks) until the blocks add up to that many seconds (at most 24 blocks), and
    their times are appended to `repeats_out` — the driver's `--steps 20` times 0.2 s, too short to say anything about spread."""
    NC, BC = len(ctxs), (batch or wl.B) // len(ctxs)       # batch: only the first `batch` sequences of the workload (the comparison legs)
    order = frame_order(wl.T, warmup + steps)
    nin = {(k, c): wl.nin(k, c * BC, (c + 1) * BC) for k in range(wl.T) for c in range(NC)}
    base = wl.data.data_ptr()

    def step(k):
        for c, cx in enumerate(ctxs):                      # asynchronous launches: the contexts' streams run concurrently
            cx.process_device(base + c * BC * wl.seq_stride + k * wl.NP * 16, wl.seq_stride, nin[(k, c)])
            if mapping:
                cx.mapping_step()

    for k in order[:warmup]:
        step(k)
    for cx in ctxs:
        cx.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if NC == 1:
        ctxs[0].profile_enable(True)                       # per-kernel hipEvents serialise nothing on one stream; with several
    t0 = time.perf_counter()                               # streams they would only measure overlapped intervals, so they stay off
    for k in order[warmup:]:
        step(k)
    for cx in ctxs:
        cx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else wl.data.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    snapshot = ctxs[0].profile() if NC == 1 else None      # the per-kernel profile is that of the K contract steps only
    if repeat_to_s > 0 and repeats_out is not None and elapsed > 0:
        cont = frame_order(wl.T, warmup + steps * 25)[warmup + steps:]          # the ping-pong replay simply goes on
        for r in range(min(24, max(0, int(np.ceil(repeat_to_s / elapsed)) - 1))):   # the same count on every rank: `elapsed` is the MAX over ranks
            if dist is not None:
                dist.barrier()
            t1 = time.perf_counter()
            for k in cont[r * steps:(r + 1) * steps]:
                step(k)
            for cx in ctxs:
                cx.synchronize()
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            if dist is not None:
                dist.barrier()
                tm = torch.tensor([el], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else wl.data.device)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                el = float(tm.item())
            repeats_out.append(el)
    if NC > 1:                                             # per-kernel profile from one extra, untimed pass on a single context
        ctxs[0].profile_enable(True)
        for k in order[warmup:]:
            ctxs[0].process_device(base + k * wl.NP * 16, wl.seq_stride, nin[(k, 0)])
            if mapping:
                ctxs[0].mapping_step()
        ctxs[0].synchronize()
    prof = snapshot if snapshot is not None else ctxs[0].profile()
    ctxs[0].profile_enable(False)
    return elapsed, prof


def roofline_of(prof, steps, B, sensor, mapping):
    """Roofline object of the dominant kernel (by accumulated hipEvent time)."""
    dname, d = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    avg_ms = d["total_ms"] / max(1, d["launches"])
    achieved = d["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    step_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in prof.values()) / max(1, steps)
    step_ms = sum(v["total_ms"] for v in prof.values()) / max(1, steps)
    r = {"kernel": dname, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 4),
         "algorithmic_bytes_per_launch": d["bytes_per_launch"],
         "whole_step": {"algorithmic_bytes": step_bytes, "kernel_ms": round(step_ms, 4),
                        "achieved_gbs": round(step_bytes / (step_ms * 1e-3) / 1e9, 2) if step_ms > 0 else 0.0,
                        "frac": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if step_ms > 0 else 0.0},
         "kernels_ms_per_step": {k: round(v["total_ms"] / steps, 4) for k, v in prof.items() if v["launches"]},
         # the same view of every profiled kernel (algorithmic bytes per launch / mean launch time against HBM peak): the dominant one changes hands between
         # k_ring_features and the planar association (two launches per step) from box to box
         "per_kernel": {k: {"avg_launch_ms": round(v["total_ms"] / v["launches"], 4), "launches_per_step": round(v["launches"] / max(1, steps), 2),
                            "achieved": round(v["bytes_per_launch"] / (v["total_ms"] / v["launches"] * 1e-3) / 1e9, 1),
                            "frac": round(v["bytes_per_launch"] / (v["total_ms"] / v["launches"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                        for k, v in prof.items() if v["launches"] and v["total_ms"] > 0}}
    # HBM traffic of the dominant kernel from the separate rocprofv3 --pmc passes of THIS command (tools/gpu_pmc.sh;
    # MI355X_MICROARCH.md: counters in their own passes; FETCH_SIZE [KiB] reports half the bytes of wide coalesced reads on
    # gfx950 -> doubled; WRITE_SIZE [KiB] as reported).  Only attached when the profiled configuration is this one.
    for name in ("pmc_traffic_latest.json", "pmc_traffic_mapping.json", "pmc_traffic_rows128.json"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if pm.get("batch") == B and pm.get("mapping") == bool(mapping) and pm.get("sensor") == sensor:
            stale = pm.get("lib_sha256") != lib_sha256()                    # counters of another build of the library (or of unknown origin) are not attached
            names = lambda dn: STAGE_KERNELS.get(dn) or [k for k in pm.get("fetch_kib", {}) if k.startswith(RK_NAMES.get(dn, dn + "<")) or k == dn][:1]
            cands = names(dname)
            if not stale and cands and all(k in pm.get("fetch_kib", {}) and k in pm.get("write_kib", {}) for k in cands):
                r["traffic"] = round(sum(2.0 * pm["fetch_kib"][k] + pm["write_kib"][k] for k in cands) * 1024.0)
                r["traffic_source"] = pm.get("source", "profiles/")
            if stale:
                r["traffic_note"] = f"profiles/{name} was taken with another build of the library (sha256 {str(pm.get('lib_sha256'))[:12]}): not attached"
            # instruction-side roofline of the kernels that are bound by VALU issue, not by HBM (the association kernels; SQ counter passes of
            # the same command): wave-instructions per second against SIMDs x clock / 4
            sq = pm.get("sq", {})
            valu = {}
            for pn in ("k_associate[plane]", "k_associate[corner]", "k_ring_features", "k_front"):
                kn = names(pn)
                if stale or pn not in prof or not prof[pn]["launches"] or not kn or kn[0] not in sq or "SQ_INSTS_VALU" not in sq[kn[0]]:
                    continue
                c, ms = sq[kn[0]], prof[pn]["total_ms"] / prof[pn]["launches"]
                g = c["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9
                valu[pn] = {"bound": "valu", "achieved": round(g, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": round(g / VALU_PEAK_GINST, 4),
                            "frac_of_fma_rate_peak": round(g / (2 * VALU_PEAK_GINST), 4), "valu_instructions_per_launch": c["SQ_INSTS_VALU"], "avg_launch_ms": round(ms, 4),
                            "busy": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (VALU_SIMDS * c["avg_us"] * 1e-6 * VALU_CLK_GHZ * 1e9), 4) if c.get("SQ_ACTIVE_INST_VALU") and c.get("avg_us") else None}
            if valu:
                r["valu"] = valu
                if dname in valu:
                    r["bound"] = "valu"     # the dominant kernel is instruction-bound: `achieved` / `frac` stay its HBM view, r["valu"][kernel] is its ceiling
                    r["binding_ceiling"] = f"roofline.valu[{dname}]: {valu[dname]['achieved']} of {VALU_PEAK_GINST} G wave-instructions/s = {valu[dname]['frac']}; achieved / peak / frac above are its algorithmic bytes against HBM"
            break
    return r


_LIB_SHA = None


def lib_sha256():
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib
        binding = importlib.import_module("a-loam_amd.binding")
        path = os.environ.get("ALOAM_MI355X_LIB", binding.LIB_PATH)
        _LIB_SHA = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
    return _LIB_SHA


def survey_b_scan(cx, wl, n_seq=8):
    """B_scan = 16 N_in + 32 N + 16 (F_c + L_c + F_s + L_s) + 2 [16 (L_c + L_s + F_c + F_s) + 360 (F_c + F_s)], with the measured
    sizes of this run (mean over a few sequences after the last timed step)."""
    binding = importlib.import_module("a-loam_amd.binding")
    n = min(n_seq, wl.B)
    L = binding.lib()
    size = lambda which: float(np.mean([L.aloam_cloud_size(cx.h, b, which) for b in range(n)]))
    N_in = float(np.mean(wl.counts[:n]))
    N, Fc, Lc, Fs, Ls = (size(w) for w in (binding.CLOUD_FULL, binding.CLOUD_SHARP, binding.CLOUD_LESS_SHARP, binding.CLOUD_FLAT, binding.CLOUD_LESS_FLAT))
    b = 16 * N_in + 32 * N + 16 * (Fc + Lc + Fs + Ls) + 2 * (16 * (Lc + Ls + Fc + Fs) + 360 * (Fc + Fs))
    return {"bytes": round(b), "N_in": round(N_in), "N": round(N), "F_c": round(Fc), "L_c": round(Lc), "F_s": round(Fs), "L_s": round(Ls)}


def host_fed(torch, binding, wl, local_rank, steps, warmup, stride=16, contexts=2):
    """The same steps with every sweep crossing PCIe: the batch sits in PINNED host memory, each step is one batched H2D copy per
    context on its copy stream into one of two device slabs, overlapped with the previous step's kernels.  `stride` = bytes per
    input record on the wire: 16 (KITTI .bin / x y z i) or 12 (x y z: the reference never reads the 4th float of its input,
    src/scanRegistration.cpp:132-133).  The batch is split over `contexts` streams so that kernels of one half overlap launch gaps
    and tails of the other (per-kernel events are off in this leg anyway)."""
    NC = contexts if wl.B % contexts == 0 and wl.B >= contexts else 1
    BC = wl.B // NC
    src = wl.data if stride == 16 else wl.data[..., :3].contiguous()
    host = src.cpu().pin_memory()
    del src
    ctxs = [wl.ctx(binding, BC, local_rank) for _ in range(NC)]
    order = frame_order(wl.T, warmup + steps)
    nin = {(k, c): wl.nin(k, c * BC, (c + 1) * BC) for k in range(wl.T) for c in range(NC)}
    hp, seq_stride = host.data_ptr(), wl.T * wl.NP * stride

    def step(k):
        for c, cx in enumerate(ctxs):
            cx.process_host(hp + c * BC * seq_stride + k * wl.NP * stride, seq_stride, nin[(k, c)], stride)

    for k in order[:warmup]:
        step(k)
    for cx in ctxs:
        cx.synchronize()
    t0 = time.perf_counter()
    for k in order[warmup:]:
        step(k)
    for cx in ctxs:
        cx.synchronize()
    el = time.perf_counter() - t0
    for cx in ctxs:
        cx.close()
    # points that cross the link per step: every context copies its rows 0 .. BC-2 with the context-wide maximum length and the last
    # row with its own (aloam_process_host)
    rows = float(sum(int(wl.counts[c * BC:(c + 1) * BC, k].max()) * (BC - 1) + int(wl.counts[(c + 1) * BC - 1, k]) for c in range(NC) for k in order[warmup:])) / steps
    del host
    return {"value": round(wl.B * steps / el, 2), "unit": "scans/s", "ms_per_step": round(1e3 * el / steps, 4), "wire_bytes_per_point": stride,
            "contexts": NC, "pcie_gbs": round(rows * stride / (el / steps) / 1e9, 2),
            "how": "pinned host batch, one strided H2D copy per context and step on a copy stream, two device slabs each (copy of step k+1 under the kernels of step k), aloam_process_host"}


def latency(binding, wl, local_rank, sweeps, mapping, graph=False):
    """Single sensor (batch 1), host buffers, blocking calls: what a drop-in ROS node gets per sweep.  graph=False: the odometry step as ~15
    separate launches instead of one hipGraph launch (ALOAM_GRAPH_MAX_BATCH=0, read by aloam_create)."""
    host = [wl.data[0, k, : wl.counts[0, k]].cpu().numpy() for k in range(wl.T)]
    prev = os.environ.get("ALOAM_GRAPH_MAX_BATCH")
    os.environ["ALOAM_GRAPH_MAX_BATCH"] = "8" if graph else "0"
    try:
        cx = wl.ctx(binding, 1, local_rank)
    finally:
        os.environ.pop("ALOAM_GRAPH_MAX_BATCH") if prev is None else os.environ.__setitem__("ALOAM_GRAPH_MAX_BATCH", prev)
    if mapping:
        cx.mapping_enable(0.4, 0.8, 262144)
    order = frame_order(wl.T, sweeps + 5)
    reg, odo, mp = [], [], []
    for i, k in enumerate(order):
        t0 = time.perf_counter()
        cx.scan_register(host[k], check=False)            # blocks until the host buffer may be reused
        t1 = time.perf_counter()
        cx.odometry_step()
        cx.synchronize()
        t2 = time.perf_counter()
        if mapping:
            cx.mapping_step()
            cx.synchronize()
        t3 = time.perf_counter()
        if i >= 5:
            reg.append(t1 - t0); odo.append(t2 - t1); mp.append(t3 - t2)
    cx.close()
    tot = np.array(reg) + np.array(odo) + np.array(mp)
    q = lambda v, p: round(1e3 * float(np.percentile(v, p)), 3)
    out = {"sweeps": len(reg), "unit": "ms per sweep", "scan_registration_median": q(reg, 50), "scan_registration_p95": q(reg, 95),
           "odometry_median": q(odo, 50), "odometry_p95": q(odo, 95), "total_median": q(tot, 50), "total_p95": q(tot, 95),
           "reference_budget_ms_per_stage": 100, "odometry_step_as_one_hipgraph_launch": bool(graph),
           "how": "batch 1, pageable host sweep -> aloam_scan_register + aloam_odometry_step + aloam_synchronize per sweep"}
    if mapping:
        out["mapping_median"], out["mapping_p95"] = q(mp, 50), q(mp, 95)
    return out


def cpu_baseline(args, wl, rank):
    """The oracle on the host cores: one thread (per-sweep median / p95), then one process per core (sequence-parallel)."""
    m = wl.model
    order = frame_order(wl.T, 2 * wl.T - 1)
    tmp = tempfile.mkdtemp(prefix="aloam_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(tmp, "sweeps.npz")
    host = wl.data[0].cpu().numpy()
    np.savez(path, T=wl.T, order=np.array(order), n_scans=m.n_scans, min_range=m.min_range, ring_from_field=int(m.ring_from_field),
             line_res=0.4, plane_res=0.8, **{f"s{k}": host[k, : wl.counts[0, k]] for k in range(wl.T)})
    worker = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), path, str(args.cpu_seconds)]
    one = json.loads(subprocess.run(worker, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    per = np.array(one["per_scan_ms"][1:])                 # the first sweep of a sequence has no odometry solve
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    procs = [subprocess.Popen(worker, stdout=subprocess.PIPE, text=True) for _ in range(cores)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    wall = time.perf_counter() - t0
    rate_all = sum(o["scans"] / o["seconds"] for o in outs)
    ref_code = reference_code_baseline(host, wl, order)
    try:
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)
    except OSError:
        pass
    return {"value": round(one["scans"] / one["seconds"], 3), "unit": "scans/s", "cores": 1, "kind": "port",
            "ms_per_scan_median": round(float(np.median(per)), 3), "ms_per_scan_p95": round(float(np.percentile(per, 95)), 3),
            "sample": f"{one['scans']} sweeps of one synthetic {wl.sensor} sequence ({args.cpu_seconds:.0f} s), oracle/ (kd-tree NN, dual-number autodiff, dense QR LM), 1 thread",
            "all_cores": {"value": round(rate_all, 2), "unit": "scans/s", "cores": cores, "kind": "port",
                          "sample": f"{cores} processes x {args.cpu_seconds:.0f} s, one independent sequence per core (same sweeps), {sum(o['scans'] for o in outs)} sweeps, wall {wall:.1f} s"},
            "reference_code": ref_code}


def reference_code_baseline(host, wl, order, sweeps=120):
    """The reference's OWN translation units (oracle/_ref: src/scanRegistration.cpp and src/laserOdometry.cpp + src/lidarFactor.hpp compiled where they lie,
    prebuilt in the container that has /root/reference; they travel to the GPU box as binaries) on the same sweeps, one thread: seconds inside the
    reference's callback / main-loop body, the drivers' file I/O excluded.  Their third-party calls (PCL VoxelGrid / KdTreeFLANN, Ceres Solve with its
    autodiff, Eigen) resolve to this repo's stand-in headers - the real libraries are not in the image - so this prices the reference's first-party code
    on stand-in third-party code, which is why `cpu_baseline.kind` stays "port" and this is reported beside it."""
    try:
        import ref_py
        if not ref_py.available():
            return None
        xs = [host[k, : wl.counts[0, k]] for k in (order * (sweeps // len(order) + 1))[:sweeps]]
        reg = ref_py.scan_registration(xs, wl.model.n_scans, wl.model.min_range, timing=True)
        ref_py.laser_odometry(reg, timing=True)
        t_reg, t_odo = ref_py.LAST_TIMING.get("scan_registration"), ref_py.LAST_TIMING.get("laser_odometry")
        if not t_reg or not t_odo:
            return None
        return {"value": round(len(xs) / (t_reg + t_odo), 3), "unit": "scans/s", "cores": 1, "kind": "reference",
                "ms_per_scan": {"scan_registration": round(1e3 * t_reg / len(xs), 3), "laser_odometry": round(1e3 * t_odo / len(xs), 3)},
                "sample": f"{len(xs)} sweeps of one synthetic {wl.sensor} sequence through oracle/_ref (the reference's scanRegistration.cpp + laserOdometry.cpp, one thread, "
                          "the two nodes one after the other); third-party calls on this repo's stand-in PCL / Ceres / Eigen headers"}
    except Exception as e:      # the baseline legs never take the bench line down
        return {"error": str(e)[:200]}


def _rotate(q, v):
    """q * v for an xyzw quaternion and an [n, 3] array (Eigen's evaluation order is irrelevant here: diagnostics only)."""
    u, w = np.asarray(q[:3], float), float(q[3])
    uv = 2.0 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def _factor_residuals(edges, planes, q, t):
    """|r| of every LidarEdgeFactor / LidarPlaneFactor at the solved pose (reference src/lidarFactor.hpp:32-40,84-87), for the
    Huber(0.1) outlier fraction."""
    re = rp = np.zeros(0)
    if len(edges):
        lp = _rotate(q, edges[:, 0:3]) + t
        a, b = edges[:, 3:6], edges[:, 6:9]
        re = np.linalg.norm(np.cross(lp - a, lp - b), axis=1) / np.linalg.norm(a - b, axis=1)
    if len(planes):
        lp = _rotate(q, planes[:, 0:3]) + t
        j, l, m = planes[:, 3:6], planes[:, 6:9], planes[:, 9:12]
        n = np.cross(j - l, j - m)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        rp = np.abs(np.sum((lp - j) * n, axis=1))
    return re, rp


def accuracy(binding, wl, local_rank, mapping=False):
    import oracle_py
    n_seq = min(2, wl.B)
    order = frame_order(wl.T, 2 * wl.T - 1)
    acc = wl.ctx(binding, n_seq, local_rank)
    gpu_poses = [[] for _ in range(n_seq)]
    base = wl.data.data_ptr()
    for k in order:
        acc.process_device(base + k * wl.NP * 16, wl.seq_stride, [int(v) for v in wl.counts[:n_seq, k]])
        for b in range(n_seq):
            gpu_poses[b].append(acc.pose(b))
    acc.close()
    host = wl.data[:n_seq].cpu().numpy()
    m = wl.model
    max_dt = max_drot = ate_sq = ate_o_sq = 0.0
    ate_n = 0
    first_err, later_err, e_out, p_out, e_med, p_med, n_e, n_p = [], [], [], [], [], [], [], []
    for b in range(n_seq):
        orc = oracle_py.Oracle(n_scans=m.n_scans, min_range=m.min_range, ring_from_field=m.ring_from_field)
        Rg, tg = wl.gt[b]
        for i, k in enumerate(order):
            orc.scan_register(host[b, k, : wl.counts[b, k]])
            po = orc.odometry_step()
            pg = gpu_poses[b][i]
            max_dt = max(max_dt, float(np.abs(po["t_lc"] - pg["t_lc"]).max()), float(np.linalg.norm(po["t_w"] - pg["t_w"])))
            max_drot = max(max_drot, quat_angle(po["q_lc"], pg["q_lc"]), quat_angle(po["q_w"], pg["q_w"]))
            if i < wl.T:   # forward part: compare with ground truth expressed in the first frame
                t_gt = Rg[0].T @ (tg[k] - tg[0])
                ate_sq += float(np.sum((pg["t_w"] - t_gt) ** 2)); ate_o_sq += float(np.sum((po["t_w"] - t_gt) ** 2)); ate_n += 1
            if 0 < i < wl.T:   # per-sweep diagnostics of the forward part: relative motion vs ground truth, factor residuals at the solution
                rel_gt = Rg[k - 1].T @ (tg[k] - tg[k - 1])
                (first_err if i == 1 else later_err).append(float(np.linalg.norm(po["t_lc"] - rel_gt)))
                eo, plo, _, _ = orc.correspondences()
                re, rp = _factor_residuals(np.asarray(eo, float), np.asarray(plo, float), po["q_lc"], po["t_lc"])
                n_e.append(len(re)); n_p.append(len(rp))
                if len(re):
                    e_out.append(float(np.mean(re > 0.1))); e_med.append(float(np.median(re)))
                if len(rp):
                    p_out.append(float(np.mean(rp > 0.1))); p_med.append(float(np.median(rp)))
    mean = lambda v: round(float(np.mean(v)), 4) if len(v) else None
    diag = {"first_solved_sweep_error_m": mean(first_err), "later_sweeps_error_m": mean(later_err),
            "edge_factors_per_sweep": mean(n_e), "plane_factors_per_sweep": mean(n_p),
            "edge_huber_outlier_fraction": mean(e_out), "plane_huber_outlier_fraction": mean(p_out),
            "edge_residual_median_m": mean(e_med), "plane_residual_median_m": mean(p_med),
            "note": "the first solved sweep starts from the identity warm start (reference src/laserOdometry.cpp:97-98) although the sensor "
                    "moves 1 m per sweep, and two outer iterations of four LM steps recover only part of it; that one sweep carries most "
                    "of the trajectory error.  Outlier = |r| > 0.1 m, the Huber(0.1) linear zone (reference src/laserOdometry.cpp:284)"}
    return {"odometry_diagnostics": diag, "gpu_vs_oracle_max_dt_m": max_dt, "gpu_vs_oracle_max_drot_rad": max_drot, "sweeps_compared": len(order) * n_seq,
            "ate_gpu_vs_gt_m": (ate_sq / max(1, ate_n)) ** 0.5, "ate_oracle_vs_gt_m": (ate_o_sq / max(1, ate_n)) ** 0.5,
            "tolerance": "1e-4 m / 1e-4 rad (BASELINE.json north_star)"}


def accuracy_mapping(binding, wl, local_rank, map_pool):
    """configs[2]: refined (scan-to-map) poses of one sequence over the stored sweeps, GPU vs oracle, and both vs the ground truth."""
    import oracle_py
    m = wl.model
    gpu = wl.ctx(binding, 1, local_rank)
    gpu.mapping_enable(0.4, 0.8, map_pool)
    orc = oracle_py.Oracle(n_scans=m.n_scans, min_range=m.min_range, ring_from_field=m.ring_from_field)
    orc.map_config(0.4, 0.8)
    host = wl.data[0].cpu().numpy()
    Rg, tg = wl.gt[0]
    base = wl.data.data_ptr()
    max_dt = max_drot = ate_g = ate_o = 0.0
    for k in range(wl.T):
        gpu.process_device(base + k * wl.NP * 16, wl.seq_stride, [int(wl.counts[0, k])])
        gpu.mapping_step()
        gpu.synchronize()
        pg = gpu.map_pose(0)
        orc.scan_register(host[k, : wl.counts[0, k]])
        po = orc.odometry_step()
        pm = orc.mapping_step(po["q_w"], po["t_w"], orc.cloud(oracle_py.CLOUD_CORNER_LAST), orc.cloud(oracle_py.CLOUD_SURF_LAST), orc.cloud(oracle_py.CLOUD_FULL))
        max_dt = max(max_dt, float(np.linalg.norm(pm["t_w"] - pg["t_w"])))
        max_drot = max(max_drot, quat_angle(pm["q_w"], pg["q_w"]))
        t_gt = Rg[0].T @ (tg[k] - tg[0])
        ate_g += float(np.sum((pg["t_w"] - t_gt) ** 2)); ate_o += float(np.sum((pm["t_w"] - t_gt) ** 2))
    gpu.close()
    return {"gpu_vs_oracle_max_dt_m": max_dt, "gpu_vs_oracle_max_drot_rad": max_drot, "sweeps_compared": wl.T,
            "ate_gpu_vs_gt_m": (ate_g / wl.T) ** 0.5, "ate_oracle_vs_gt_m": (ate_o / wl.T) ** 0.5, "tolerance": "1e-4 m / 1e-4 rad (BASELINE.json north_star)"}


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    if os.environ.get(RANK_ENV_ONLY):
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")} | {"argv": sys.argv[1:]}), flush=True)
        return
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    shared = bool(os.environ.get(SHARED_GPU_ENV))
    if shared:
        local_rank %= torch.cuda.device_count()
    if torch.cuda.device_count() <= local_rank:            # before RCCL is asked for a device that does not exist
        raise SystemExit(f"bench.py rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP device(s) are visible: start one rank per GPU "
                         f"(--nproc-per-node <= {torch.cuda.device_count()}), or set {SHARED_GPU_ENV}=1 to let ranks share a device (test hook, gloo control plane)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get(FORCE_DIST_ENV):       # (test hook: the control plane also with a single rank, e.g. RCCL on the one GPU of the box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")                # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=dev) # control plane only: barrier + MAX of the elapsed time

    binding = importlib.import_module("a-loam_amd.binding")
    syn = importlib.import_module("a-loam_amd.synthetic")
    if not os.path.exists(binding.LIB_PATH):
        binding.build()

    B, T = args.batch, args.frames
    free0, total_hbm = torch.cuda.mem_get_info(dev)
    if args.mapping:        # configs[2] as the workload of this run (rocprofv3 / counter passes): the travelling sensor at steady-state map depth
        assert args.sensor == "HDL-64"
        B, T, args.warmup = args.map_batch, args.map_frames, max(args.warmup, args.map_warmup)
        wl = TravelWorkload(syn, torch, B, T, rank, dev)
    elif args.travel:
        assert args.sensor == "HDL-64"
        wl = TravelWorkload(syn, torch, B, T, rank, dev)
    else:
        wl = Workload(syn, torch, args.sensor, B, T, rank, dev, rough=args.rough)
    NC = max(1, args.contexts)
    assert B % NC == 0, "--batch must be a multiple of --contexts"
    ctxs = [wl.ctx(binding, B // NC, local_rank) for _ in range(NC)]
    if args.reference_order:
        for c in ctxs:
            c.set_voxel_sum_order(True)
    if args.mapping:
        for c in ctxs:
            c.mapping_enable(0.4, 0.8, args.map_pool)      # launch/aloam_velodyne_HDL_64.launch: mapping_line / plane_resolution
    more = []
    # (not under --no-extras: that is what the rocprofv3 / counter passes and the A/B scripts run, and they want exactly W + K steps)
    elapsed, prof = timed_resident(torch, dist, world, ctxs, wl, args.steps, args.warmup, args.mapping, repeat_to_s=0.0 if args.no_extras else args.repeat_to_seconds, repeats_out=more)
    bscan = survey_b_scan(ctxs[0], wl)
    mstate = map_state(ctxs[0]) if args.mapping else None
    free1, _ = torch.cuda.mem_get_info(dev)                # inputs + contexts of THIS rank (of every rank that shares the device under the test hook)
    try:
        import psutil
        rss, host_ram = psutil.Process().memory_info().rss, psutil.virtual_memory().total
    except ImportError:
        rss = host_ram = None
    for cx in ctxs:
        cx.close()
    value = world * B * args.steps / elapsed

    out = {"metric": "HDL-64 scans/sec (whole node)", "value": round(value, 2), "unit": "scans/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 features / f64 solve", "data": "synthetic",
           "config": {"workload": wl.describe(args.mapping), "sequences_per_gpu": B, "contexts_per_gpu": NC, "stored_frames": T, "voxel_sum_order": "reference (std::sort replayed, validation mode)" if args.reference_order else "input order (throughput path)",
                      "points_per_sweep": wl.NP, "parallelism": f"{world} x independent sequence shards, no collectives"},
           "roofline": roofline_of(prof, args.steps, B, args.sensor, args.mapping), "input_generation_s": round(wl.gen_s, 2),
           "timed_region_s": round(elapsed, 3), "library_sha256": lib_sha256(),
           "memory": {"hbm_used_bytes": int(free0 - free1), "hbm_bytes_per_sequence": int((free0 - free1) / max(1, B)), "hbm_total_bytes": int(total_hbm),
                      "host_rss_bytes": rss, "host_ram_bytes": host_ram, "input_bytes": int(B * T * wl.NP * 16)}}
    if mstate is not None:
        out["map_state"] = mstate
    if more:     # spread of identical K-step blocks timed right after the contract block (which alone is `value` / `ms_per_step`)
        blocks = [elapsed] + more
        ms = sorted(1e3 * x / args.steps for x in blocks)
        out["repeated_blocks"] = {"blocks": len(blocks), "steps_each": args.steps, "total_timed_s": round(sum(blocks), 3),
                                  "ms_per_step": {"first": round(1e3 * elapsed / args.steps, 4), "median": round(float(np.median(ms)), 4), "min": round(ms[0], 4), "max": round(ms[-1], 4)},
                                  "value_median": round(world * B * args.steps / (float(np.median(blocks))), 2)}
    # whole-path view with SURVEY.md 8(d)'s B_scan (one figure per sweep, measured sizes) next to the sum of the per-kernel floors above
    out["roofline"]["survey_b_scan"] = dict(bscan, achieved_gbs=round(value / world * bscan["bytes"] / 1e9, 2), frac=round(value / world * bscan["bytes"] / 1e9 / HBM_PEAK_GBS, 5),
                                            frac_of_measured_copy_peak=round(value / world * bscan["bytes"] / 6.29e12, 5))

    extras = rank == 0 and world == 1 and not args.no_extras
    xsteps = min(args.steps, args.extra_steps)            # steps of the secondary legs (host-fed, four contexts, configs[2] / [3])
    if extras and NC == 1 and B % 4 == 0:
        # the same steps with the batch split over four contexts (= four HIP streams): tails and launch gaps of one quarter are
        # filled by the others.  Per-kernel events are off in this leg (overlapped intervals do not add up), which is why it is
        # not the headline: `value` and `roofline` above come from one context with events on its own stream.
        c4 = [wl.ctx(binding, B // 4, local_rank) for _ in range(4)]
        el4, _ = timed_resident(torch, dist, 1, c4, wl, xsteps, args.warmup, False)
        for cx in c4:
            cx.close()
        out["overlapped_streams"] = {"contexts": 4, "value": round(B * xsteps / el4, 2), "unit": "scans/s", "ms_per_step": round(1e3 * el4 / xsteps, 4), "steps": xsteps}
    if extras:
        hf = host_fed(torch, binding, wl, local_rank, xsteps, args.warmup, stride=16)
        out["value_host_input"] = hf["value"]              # per GPU, PCIe-inclusive, the 16-byte wire format; never the headline
        out["host_input"] = hf
        out["host_input_xyz12"] = host_fed(torch, binding, wl, local_rank, xsteps, args.warmup, stride=12)
        out["latency"] = latency(binding, wl, local_rank, args.latency_sweeps, mapping=False)
        out["latency"]["with_hipgraph_odometry_step"] = {k: v for k, v in latency(binding, wl, local_rank, args.latency_sweeps, mapping=False, graph=True).items() if k.endswith(("median", "p95"))}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["accuracy"] = accuracy(binding, wl, local_rank)
        if not args.rough:     # the same legs on KITTI-shaped irregular sweeps (dropouts, ragged rings, noisy sectors, repeated returns)
            wlr = Workload(syn, torch, args.sensor, 2, T, rank, dev, rough=True)
            out["accuracy_rough"] = accuracy(binding, wlr, local_rank)
            if args.sensor == "HDL-64":
                out["accuracy_rough"]["with_mapping"] = accuracy_mapping(binding, wlr, local_rank, args.map_pool)
            del wlr
        out["cpu_baseline"] = cpu_baseline(args, wl, rank)
    if extras and not args.mapping and args.sensor == "HDL-64":
        # ---- BASELINE.json configs[2]: the same sweeps with the scan-to-map refinement after every sweep
        steps2, warm2 = max(4, min(args.steps, args.extra_steps)), args.warmup
        Bx = min(B, 1024)                                  # the comparison legs keep the batch of the earlier rounds
        cx = wl.ctx(binding, Bx, local_rank)
        cx.mapping_enable(0.4, 0.8, args.map_pool)
        el2, prof2 = timed_resident(torch, dist, 1, [cx], wl, steps2, warm2, True, batch=Bx)
        info = cx.map_info(0)
        cx.close()
        lat_map = latency(binding, wl, local_rank, max(30, args.latency_sweeps // 3), mapping=True)
        acc_map = accuracy_mapping(binding, wl, local_rank, args.map_pool) if not args.no_cpu_baseline else None
        out["workloads"] = {"configs[2] odometry + laserMapping": {
            "workload": wl.describe(True), "value": round(Bx * steps2 / el2, 2), "unit": "scans/s", "ms_per_step": round(1e3 * el2 / steps2, 4),
            "steps": steps2, "warmup": warm2, "sequences_per_gpu": Bx, "map_pool_points": args.map_pool,
            "roofline": roofline_of(prof2, steps2, Bx, "HDL-64", True), "map_state_seq0": {k: info[k] for k in ("frame_count", "from_map_corner", "from_map_surf", "corner_stack", "surf_stack")},
            "latency": lat_map}}
        if acc_map is not None:
            out["workloads"]["configs[2] odometry + laserMapping"]["accuracy"] = acc_map
        out["workloads"]["configs[2] odometry + laserMapping"]["note"] = "the round-5 workload (six sweeps on a 30 m circle, ping-pong): a SHALLOW submap; kept for comparison with earlier rounds"
        del wl
        torch.cuda.empty_cache()
        # ---- the same at steady-state map depth: a travelling sensor, 100 distinct sweeps per sequence, submap stationary after the warm-up
        Bt = args.map_batch
        wlt = TravelWorkload(syn, torch, Bt, args.map_frames, rank, dev)
        cx = wlt.ctx(binding, Bt, local_rank)
        cx.mapping_enable(0.4, 0.8, args.map_pool)
        steps_t = max(steps2, 60)
        elt, proft = timed_resident(torch, dist, 1, [cx], wlt, steps_t, args.map_warmup, True)
        out["workloads"]["configs[2] odometry + laserMapping, steady-state map depth"] = {
            "workload": wlt.describe(True), "value": round(Bt * steps_t / elt, 2), "unit": "scans/s", "ms_per_step": round(1e3 * elt / steps_t, 4),
            "steps": steps_t, "warmup": args.map_warmup, "sequences_per_gpu": Bt, "map_pool_points_at_start": args.map_pool,
            "roofline": roofline_of(proft, steps_t, Bt, "HDL-64", True), "map_state": map_state(cx), "input_generation_s": round(wlt.gen_s, 2),
            "ms_per_1024_sequences": round(1e3 * elt / steps_t * 1024 / Bt, 3)}
        cx.close()
        del wlt
        torch.cuda.empty_cache()
        # ---- BASELINE.json configs[3]: 128 rings x 2048 columns stress (ring index from the 4th float)
        T3 = min(T, 4)
        wl3 = Workload(syn, torch, "ROWS128", Bx, T3, rank, dev)
        cx = wl3.ctx(binding, Bx, local_rank)
        el3, prof3 = timed_resident(torch, dist, 1, [cx], wl3, steps2, warm2, False)
        cx.close()
        out["workloads"]["configs[3] 128x2048 stress"] = {
            "workload": wl3.describe(False), "value": round(Bx * steps2 / el3, 2), "unit": "scans/s", "ms_per_step": round(1e3 * el3 / steps2, 4),
            "steps": steps2, "warmup": warm2, "sequences_per_gpu": Bx, "points_per_sweep": wl3.NP,
            "roofline": roofline_of(prof3, steps2, Bx, "ROWS128", False), "input_generation_s": round(wl3.gen_s, 2)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
