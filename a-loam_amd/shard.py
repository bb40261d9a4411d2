"""Multi-GPU sharding helpers: sequences are independent units, one process per GPU, NO data-path
collective.  The only communication is control-plane: a barrier around the timed region, a MAX-reduction of the elapsed
time, and (optionally) a gather of the per-sequence poses to rank 0.  Works with the `nccl` (= RCCL) backend on GPUs
and with `gloo` on CPU (tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_sequences(n_sequences: int, rank: int, world: int) -> list[int]:
    """Global sequence ids owned by `rank`: sequence s -> rank s mod world (round robin)."""
    return list(range(rank, n_sequences, world))


def global_sequence_id(local_index: int, rank: int, world: int) -> int:
    return rank + local_index * world


def max_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_poses(local_ids: list[int], local_poses: torch.Tensor, n_sequences: int, device="cpu"):
    """local_poses [n_local, 7] (t_w xyz, q_w xyzw) -> [n_sequences, 7] on every rank, ordered by global id.
    Control-plane only (a few hundred bytes per sequence); not part of the timed data path."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    out = torch.zeros((n_sequences, 7), dtype=torch.float64, device=device)
    if world == 1:
        out[torch.tensor(local_ids, dtype=torch.long)] = local_poses.to(device=device, dtype=torch.float64)
        return out
    n_max = (n_sequences + world - 1) // world
    pad = torch.zeros((n_max, 8), dtype=torch.float64, device=device)
    pad[: len(local_ids), :7] = local_poses.to(device=device, dtype=torch.float64)
    pad[: len(local_ids), 7] = torch.tensor(local_ids, dtype=torch.float64, device=device) + 1.0
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    for bfr in bufs:
        ids = bfr[:, 7].long() - 1
        keep = ids >= 0
        out[ids[keep]] = bfr[keep, :7]
    return out
