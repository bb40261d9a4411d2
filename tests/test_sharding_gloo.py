"""world_size-2 `gloo` test of the N>1 path: sequences shard round-robin with no data-path collective; only the
barrier, the MAX-reduction of the timed region and the (control-plane) pose gather touch the process group."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_seq, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("a-loam_amd.shard")
    syn = importlib.import_module("a-loam_amd.synthetic")
    import oracle_py as O
    ids = shard.shard_sequences(n_seq, rank, world)
    poses = []
    for gid in ids:                                   # each rank processes ONLY its own sequences, nothing is exchanged
        scans, R, t, model = syn.make_sequence("VLP-16", 2, seed=50 + gid, columns=360)
        orc = O.Oracle(n_scans=16, min_range=model.min_range)
        for s in scans:
            orc.scan_register(s.numpy()); p = orc.odometry_step()
        poses.append(np.r_[p["t_w"], p["q_w"]])
    dist.barrier()
    elapsed = shard.max_over_ranks(1.0 + rank)        # the slowest rank defines the step time
    allp = shard.gather_poses(ids, torch.tensor(np.array(poses)), n_seq)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.r_[elapsed, allp.numpy().ravel()])
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    shard = importlib.import_module("a-loam_amd.shard")
    n_seq, world = 5, 2
    parts = [shard.shard_sequences(n_seq, r, world) for r in range(world)]
    assert sorted(sum(parts, [])) == list(range(n_seq)) and not set(parts[0]) & set(parts[1])
    assert all(shard.global_sequence_id(i, r, world) == g for r in range(world) for i, g in enumerate(parts[r]))
    mp.spawn(_worker, args=(world, _free_port(), n_seq, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert r0[0] == 2.0 and r1[0] == 2.0                                  # MAX over ranks
    assert np.array_equal(r0[1:], r1[1:])                                 # every rank sees the same gathered table
    table = r0[1:].reshape(n_seq, 7)
    assert np.all(np.linalg.norm(table[:, :3], axis=1) > 0.5)             # every sequence was processed by exactly one rank
    assert np.allclose(np.linalg.norm(table[:, 3:], axis=1), 1.0, atol=1e-9)
