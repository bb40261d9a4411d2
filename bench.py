#!/usr/bin/env python
"""bench.py — HDL-64 scans/sec of the MI355X A-LOAM hot path (scan registration + scan-to-scan odometry).

One "step" = one sweep of every one of `--batch` independent sequences per GPU pushed through
aloam_process_device(): stage 1 (reference src/scanRegistration.cpp:127-411) + stage 2 (reference
src/laserOdometry.cpp:265-506,554-568).  Workload = BASELINE.json configs[1] with synthetic data (KITTI is not in
the image): 64 rings x 2048 columns = 131072 points per sweep, ring-major like KITTI .bin files, minimum_range 5,
noise sigma 0.02 m, sensor driven 1 m per sweep on a 30 m circle.  Inputs are resident in HBM before the timed
region; every sequence replays its `--frames` stored sweeps forwards and backwards (consecutive sweeps are always
1 m apart, so the odometry problem is the real one at every step).

Multi-GPU (--gpus N under torch.distributed.run): sequences are independent, each rank owns `--batch` of them, no
data-path collective; a barrier brackets the timed region and the max time over ranks is used ("scaling": "weak").

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  roofline     - dominant kernel, algorithmic bytes per launch / its mean hipEvent duration, vs 8 TB/s HBM
  cpu_baseline - the CPU oracle (oracle/, a restatement of the reference; "kind": "port") timed on one host core on
                 the first sequence's sweeps, in the same run
  accuracy     - per-sweep pose difference GPU vs oracle (parity) and trajectory ATE vs the synthetic ground truth
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s achievable float4 copy)


def quat_angle(qa, qb):
    """Rotation angle between two xyzw quaternions."""
    d = abs(float(np.dot(qa, qb))) / (np.linalg.norm(qa) * np.linalg.norm(qb))
    return 2.0 * float(np.arccos(min(1.0, d)))


def rot_to_quat(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_quat()


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="independent sequences per GPU")
    ap.add_argument("--frames", type=int, default=6, help="stored sweeps per sequence (replayed ping-pong)")
    ap.add_argument("--sensor", default="HDL-64")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mapping", action="store_true", help="BASELINE configs[2]: also run the scan-to-map refinement every sweep")
    ap.add_argument("--map-pool", type=int, default=262144, help="device map capacity per sequence and class (points)")
    ap.add_argument("--contexts", type=int, default=1, help="split the batch over this many contexts (= HIP streams) on the same GPU so that "
                    "kernels with different bottlenecks overlap")
    ap.add_argument("--host-input", action="store_true", help="feed the sweeps from host memory through aloam_scan_register (PCIe-inclusive rate; "
                    "reported as value_host_input next to the HBM-resident value, never instead of it)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    binding = importlib.import_module("a-loam_amd.binding")
    syn = importlib.import_module("a-loam_amd.synthetic")
    if not os.path.exists(binding.LIB_PATH):
        binding.build()

    B, T = args.batch, args.frames
    model = syn.sensor_model(args.sensor, device=dev)
    NP = model.dirs.shape[0]
    # ---- synthetic input, generated on the GPU, resident in HBM: [B][T][NP][4] float32 ----
    t_gen = time.time()
    data = torch.zeros((B, T, NP, 4), dtype=torch.float32, device=dev)
    counts = np.zeros((B, T), np.int32)
    worlds = [syn.make_world(100 + w).to(dev) for w in range(8)]
    gt = {}
    for b in range(B):
        gseq = rank * B + b
        R, tt = syn.trajectory(T, step=1.0, seed=gseq, start_angle=0.37 * gseq)
        gen = torch.Generator(device=dev).manual_seed(9000 + gseq)
        for k in range(T):
            s = syn.render_scan(worlds[gseq % len(worlds)], model, R[k], tt[k], 0.02, gen)
            counts[b, k] = s.shape[0]
            data[b, k, : s.shape[0]] = s
        if b < 4:
            gt[b] = (R.numpy(), tt.numpy())
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    NC = max(1, args.contexts)
    assert B % NC == 0, "--batch must be a multiple of --contexts"
    BC = B // NC
    ctxs = [binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field, batch=BC,
                          max_points=NP, max_ring_points=2059 if model.columns <= 2048 else 4107, device=local_rank) for _ in range(NC)]
    ctx = ctxs[0]
    if args.mapping:
        for c in ctxs:
            c.mapping_enable(0.4, 0.8, args.map_pool)      # launch/aloam_velodyne_HDL_64.launch: mapping_line / plane_resolution
    seq_stride = T * NP * 16
    order = frame_order(T, args.warmup + args.steps)
    nin = {(k, c): (ctypes.c_int * BC)(*[int(v) for v in counts[c * BC:(c + 1) * BC, k]]) for k in range(T) for c in range(NC)}
    base = data.data_ptr()

    def step(k):
        for c, cx in enumerate(ctxs):                      # asynchronous launches: the contexts' streams run concurrently
            cx.process_device(base + c * BC * seq_stride + k * NP * 16, seq_stride, nin[(k, c)])
            if args.mapping:
                cx.mapping_step()

    for k in order[: args.warmup]:
        step(k)
    for cx in ctxs:
        cx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if NC == 1:
        ctx.profile_enable(True)                           # per-kernel hipEvents serialise nothing on one stream; with several
    t0 = time.perf_counter()                               # streams they would only measure overlapped intervals, so they stay off
    for k in order[args.warmup:]:
        step(k)
    for cx in ctxs:
        cx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if NC > 1:                                             # per-kernel profile from one extra, untimed pass on a single context
        ctx.profile_enable(True)
        for k in order[args.warmup:]:
            ctx.process_device(base + k * NP * 16, seq_stride, nin[(k, 0)])
            if args.mapping:
                ctx.mapping_step()
        ctx.synchronize()
    prof = ctx.profile()
    ctx.profile_enable(False)

    host_rate = None
    if args.host_input:                                    # same steps, inputs handed over as host buffers (one H2D copy per sweep)
        host = data.cpu().numpy()
        hctx = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field, batch=B,
                             max_points=NP, max_ring_points=2059 if model.columns <= 2048 else 4107, device=local_rank)
        def hstep(k):
            hctx.scan_register([host[b, k, : counts[b, k]] for b in range(B)], check=False)
            hctx.odometry_step()
        for k in order[: args.warmup]:
            hstep(k)
        hctx.synchronize()
        th = time.perf_counter()
        for k in order[args.warmup:]:
            hstep(k)
        hctx.synchronize()
        host_rate = B * args.steps / (time.perf_counter() - th)
        hctx.close()

    total_scans = world * B * args.steps
    value = total_scans / elapsed

    # ---- roofline object for the dominant kernel ----
    dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    dname, d = dom
    avg_ms = d["total_ms"] / max(1, d["launches"])
    achieved = d["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {"kernel": dname, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": d["bytes_per_launch"],
                "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in prof.items()}}

    # HBM traffic of the dominant kernel from the separate rocprofv3 --pmc passes of THIS command (tools/gpu_round.sh;
    # MI355X_MICROARCH.md: counters in their own passes; FETCH_SIZE [KiB] reports half the bytes of wide coalesced reads on
    # gfx950 -> doubled; WRITE_SIZE [KiB] as reported).  Only attached when the profiled configuration is this one.
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
        if pm.get("batch") == B and pm.get("mapping") == bool(args.mapping) and pm.get("sensor") == args.sensor:
            rk = {"k_associate[plane]": "k_associate<true, false>", "k_associate[corner]": "k_associate<false, false>", "k_ring_features": "k_ring_features<2048>",
                  "k_solve": "k_solve<false>"}.get(dname, dname)
            if rk not in pm["fetch_kib"]:
                rk = rk.replace(", false>", ">").replace("<false>", "")          # names of builds before the de-skew template parameter
            if rk in pm["fetch_kib"] and rk in pm["write_kib"]:
                roofline["traffic"] = round((2.0 * pm["fetch_kib"][rk] + pm["write_kib"][rk]) * 1024.0)
                roofline["traffic_source"] = pm.get("source", "profiles/")
    except (OSError, ValueError, KeyError):
        pass

    out = {"metric": "HDL-64 scans/sec (whole node)", "value": round(value, 2), "unit": "scans/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 features / f64 solve", "data": "synthetic",
           "config": {"workload": f"synthetic {args.sensor} {model.n_scans}x{model.columns} ({NP} pts/sweep), " + ("odometry + laserMapping scan-to-map refinement every sweep" if args.mapping else "odometry only (scan registration + scan-to-scan odometry, no laserMapping)"),
                      "sequences_per_gpu": B, "contexts_per_gpu": NC, "stored_frames": T, "points_per_sweep": NP, "parallelism": f"{world} x independent sequence shards, no collectives"},
           "roofline": roofline, "input_generation_s": round(t_gen, 2)}
    if host_rate is not None:
        out["value_host_input"] = round(host_rate, 2)      # per rank, PCIe-inclusive, pageable host buffers

    # ---- CPU baseline + accuracy (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_py
        n_acc_seq = min(2, B)
        acc_order = frame_order(T, 2 * T - 1)
        acc = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field, batch=n_acc_seq,
                            max_points=NP, max_ring_points=2059 if model.columns <= 2048 else 4107, device=local_rank)
        gpu_poses = [[] for _ in range(n_acc_seq)]
        for k in acc_order:
            acc.process_device(base + k * NP * 16, seq_stride, [int(v) for v in counts[:n_acc_seq, k]])
            for b in range(n_acc_seq):
                gpu_poses[b].append(acc.pose(b))
        acc.close()
        host = data[:n_acc_seq].cpu().numpy()
        cpu_t, cpu_scans = 0.0, 0
        max_dt, max_drot, ate_sq, ate_n, ate_o_sq = 0.0, 0.0, 0.0, 0, 0.0
        for b in range(n_acc_seq):
            orc = oracle_py.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
            Rg, tg = gt[b]
            for i, k in enumerate(acc_order):
                x = host[b, k, : counts[b, k]]
                t1 = time.perf_counter()
                orc.scan_register(x)
                po = orc.odometry_step()
                cpu_t += time.perf_counter() - t1
                cpu_scans += 1
                pg = gpu_poses[b][i]
                max_dt = max(max_dt, float(np.abs(po["t_lc"] - pg["t_lc"]).max()), float(np.linalg.norm(po["t_w"] - pg["t_w"])))
                max_drot = max(max_drot, quat_angle(po["q_lc"], pg["q_lc"]), quat_angle(po["q_w"], pg["q_w"]))
                if i < T:   # forward part: compare with ground truth expressed in the first frame
                    t_gt = Rg[0].T @ (tg[k] - tg[0])
                    ate_sq += float(np.sum((pg["t_w"] - t_gt) ** 2)); ate_o_sq += float(np.sum((po["t_w"] - t_gt) ** 2)); ate_n += 1
            if cpu_t > args.cpu_seconds and b + 1 < n_acc_seq:
                break
        # keep timing the oracle on further sweeps until the budget is used
        orc = oracle_py.Oracle(n_scans=model.n_scans, min_range=model.min_range, ring_from_field=model.ring_from_field)
        i = 0
        while cpu_t < args.cpu_seconds:
            k = acc_order[i % len(acc_order)]; i += 1
            x = host[0, k, : counts[0, k]]
            t1 = time.perf_counter(); orc.scan_register(x); orc.odometry_step(); cpu_t += time.perf_counter() - t1; cpu_scans += 1
        out["cpu_baseline"] = {"value": round(cpu_scans / cpu_t, 3), "unit": "scans/s", "cores": 1, "kind": "port",
                               "sample": f"{cpu_scans} sweeps of the same synthetic {args.sensor} sequences, oracle/ (kd-tree NN, dual-number autodiff, dense QR LM), 1 thread, host cores available: {os.cpu_count()}"}
        out["accuracy"] = {"gpu_vs_oracle_max_dt_m": max_dt, "gpu_vs_oracle_max_drot_rad": max_drot, "sweeps_compared": len(acc_order) * n_acc_seq,
                           "ate_gpu_vs_gt_m": (ate_sq / max(1, ate_n)) ** 0.5, "ate_oracle_vs_gt_m": (ate_o_sq / max(1, ate_n)) ** 0.5,
                           "tolerance": "1e-4 m / 1e-4 rad (BASELINE.json north_star)"}
    for cx in ctxs:
        cx.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
