"""Host-side logic of bench.py that needs no GPU: the ranks `--gpus N` starts itself, the replay order, the CPU-baseline worker."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_spawn_environment_is_what_a_launcher_would_export():
    b = _bench()
    envs = b.rank_envs(4, 29511, base={"PATH": "/x"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/x" for e in envs)


def test_gpus_flag_is_read():
    """Round 1's bench parsed --gpus and never used it: `python bench.py --gpus 8` ran one rank."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.gpus > 1" in src and "self_spawn(args)" in src and '"n_gpus": world' in src


def test_replay_order_keeps_consecutive_sweeps_adjacent():
    b = _bench()
    o = b.frame_order(6, 23)
    assert o[:12] == [0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0, 1] and all(abs(x - y) == 1 for x, y in zip(o, o[1:]))
    assert b.frame_order(1, 3) == [0, 0, 0]


def test_cpu_baseline_worker(tmp_path, syn):
    scans, R, t, model = syn.make_sequence("VLP-16", 3, seed=2, columns=300)
    path = str(tmp_path / "s.npz")
    np.savez(path, T=3, order=np.array([0, 1, 2, 1, 0]), n_scans=16, min_range=model.min_range, ring_from_field=0, line_res=0.2, plane_res=0.4,
             **{f"s{k}": s.numpy() for k, s in enumerate(scans)})
    for extra in ([], ["--mapping"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), path, "0.3"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        o = json.loads(r.stdout.strip().splitlines()[-1])
        assert o["scans"] >= 3 and len(o["per_scan_ms"]) == o["scans"] and o["seconds"] > 0


def test_self_spawn_really_starts_the_ranks():
    """`python bench.py --gpus 3` executed for real: the parent starts three children which (test hook ALOAM_BENCH_RANK_ENV_ONLY)
    print the environment they were given and exit; no GPU and no torch involved."""
    env = dict(os.environ, ALOAM_BENCH_RANK_ENV_ONLY="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--batch", "7", "--steps", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ranks = sorted((json.loads(line) for line in r.stdout.strip().splitlines()), key=lambda d: int(d["RANK"]))
    assert [d["RANK"] for d in ranks] == ["0", "1", "2"] and [d["LOCAL_RANK"] for d in ranks] == ["0", "1", "2"]
    assert all(d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for d in ranks)
    assert len({d["MASTER_PORT"] for d in ranks}) == 1 and int(ranks[0]["MASTER_PORT"]) > 0
    assert all(d["argv"] == ["--gpus", "3", "--batch", "7", "--steps", "2"] for d in ranks)      # every rank gets the caller's flags


def test_roofline_objects_attach_only_counters_of_this_library(tmp_path, monkeypatch):
    """bench.py's roofline object: dominant kernel by hipEvent time, HBM view from the algorithmic bytes, `traffic` and the instruction-side
    `valu` objects from the counter set under profiles/ — but only when that set carries the sha256 of the library the bench is running."""
    b = _bench()
    prof = {"k_ring_features": {"total_ms": 24.0, "launches": 10, "bytes_per_launch": 2.4e9},
            "k_associate[plane]": {"total_ms": 22.0, "launches": 20, "bytes_per_launch": 0.74e9},
            "k_scatter": {"total_ms": 8.0, "launches": 10, "bytes_per_launch": 4.0e9}}
    pm = {"batch": 1024, "mapping": False, "sensor": "HDL-64", "lib_sha256": "abc", "source": "test",
          "fetch_kib": {"k_ring_features<2048>": 1.0e6, "k_associate_pair<true, false>": 9.0e5}, "write_kib": {"k_ring_features<2048>": 5.0e5, "k_associate_pair<true, false>": 4.0e4},
          "sq": {"k_ring_features<2048>": {"SQ_INSTS_VALU": 1.0e9, "SQ_ACTIVE_INST_VALU": 1.0e9, "avg_us": 2400.0},
                 "k_associate_pair<true, false>": {"SQ_INSTS_VALU": 4.6e8, "SQ_ACTIVE_INST_VALU": 4.6e8, "avg_us": 1000.0}}}
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    json.dump(pm, open(root / "profiles" / "pmc_traffic_latest.json", "w"))
    monkeypatch.setattr(b, "ROOT", str(root))
    monkeypatch.setattr(b, "lib_sha256", lambda: "abc")
    r = b.roofline_of(prof, 10, 1024, "HDL-64", False)
    assert r["kernel"] == "k_ring_features" and abs(r["achieved"] - 1000.0) < 1e-6 and abs(r["frac"] - 0.125) < 1e-6
    assert r["traffic"] == round((2 * 1.0e6 + 5.0e5) * 1024)
    v = r["valu"]["k_ring_features"]
    assert r["bound"] == "valu" and v["bound"] == "valu" and abs(v["achieved"] - 1.0e9 / 2.4e-3 / 1e9) < 0.1 and abs(v["peak"] - 614.4) < 1e-9
    assert abs(v["busy"] - 1.0e9 * 4 / (1024 * 2.4e-3 * 2.4e9)) < 1e-3 and "binding_ceiling" in r
    assert abs(r["valu"]["k_associate[plane]"]["valu_instructions_per_launch"] - 4.6e8) < 1
    monkeypatch.setattr(b, "lib_sha256", lambda: "another build")
    r = b.roofline_of(prof, 10, 1024, "HDL-64", False)
    assert r["traffic"] is None and "valu" not in r and r["bound"] == "hbm" and "not attached" in r["traffic_note"]
