"""Multi-GPU plumbing: environments shard across ranks with NO data-path collective (they are independent,
experiments/eval_policy_parallel.py:266-280); ranks meet only to agree on the wall time (MAX) and to all-gather one
fixed-size result record each.  Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def env_shard(n_total: int, rank: int, world: int):
    """Contiguous split of `n_total` environments (e.g. 256 envs / 8 GPUs -> 32 each); remainder to the low ranks."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(record, device):
    """All-gather one fixed-size float64 record per rank -> tensor [world, len(record)] on every rank."""
    rec = torch.as_tensor(record, dtype=torch.float64, device=device).reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec[None].clone()
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out)


def throughput(records: torch.Tensor, elapsed_max_s: float) -> float:
    """Whole-job env-steps/s: sum over ranks of envs * steps, divided by the slowest rank's time."""
    return float((records[:, 0] * records[:, 1]).sum().item() / elapsed_max_s)
