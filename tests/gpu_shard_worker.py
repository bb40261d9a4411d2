"""Worker of tests/test_gpu_parity.py::test_two_ranks_shard_sequences_on_the_hip_path: one rank of a two-rank job (gloo control plane, the
ranks share the one GPU of the box).  Each rank pushes ONLY its own sequences (sequence s -> rank s mod world, a-loam_amd/shard.py) through
its own HIP context; nothing but the final pose table crosses the process group."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_seq, frames, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    binding = importlib.import_module("a-loam_amd.binding")
    shard = importlib.import_module("a-loam_amd.shard")
    syn = importlib.import_module("a-loam_amd.synthetic")
    ids = shard.shard_sequences(n_seq, rank, world)
    seqs = [syn.make_sequence("VLP-16", frames, seed=70 + g, columns=600) for g in ids]
    model = seqs[0][3]
    gpu = binding.Aloam(n_scans=model.n_scans, min_range=model.min_range, batch=len(ids), max_points=16 * 600 + 64, device=0)
    for k in range(frames):
        gpu.scan_register([s[0][k].numpy() for s in seqs])
        gpu.odometry_step()
    gpu.synchronize()
    poses = np.array([np.r_[gpu.pose(b)["t_w"], gpu.pose(b)["q_w"]] for b in range(len(ids))])
    gpu.close()
    dist.barrier()
    table = shard.gather_poses(ids, torch.tensor(poses), n_seq)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), table.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
