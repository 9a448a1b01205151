"""Multi-GPU plumbing: environments shard across ranks with NO data-path collective (they are independent,
experiments/eval_policy_parallel.py:266-280); ranks meet only to agree on the wall time (MAX) and to all-gather one
fixed-size result record each.  Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n_ranks: int, script: str, argv, port: int | None = None):
    """The command line that starts ``script`` as ``n_ranks`` processes of one node, one per GPU — what the reference's
    multi-GPU entry does by spawning one worker per device (experiments/eval_policy_parallel.py:266-280), through
    torch.distributed.run so that RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* reach every worker."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script, *[str(a) for a in argv]]


def self_launch(n_ranks: int, script: str, argv, env=None, timeout=None) -> int:
    """Run ``script`` as ``n_ranks`` ranks and return the launcher's exit code.  Called by a script that was started as
    a single plain process with ``--gpus N`` (N > 1)."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    e["R2S_SELF_LAUNCHED"] = "1"
    return subprocess.call(launch_command(n_ranks, script, argv), env=e, timeout=timeout)


def rank_info():
    """(rank, local_rank, world) from the environment torch.distributed.run sets up; (0, 0, 1) for a plain process."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def resolve_world(gpus_arg: int, script: str, argv):
    """Reconcile ``--gpus N`` with how the process was started.  Returns (rank, local_rank, world) for a rank that should
    run the workload, or exits: a plain process asked for N > 1 re-launches itself as N ranks and exits with their code; a
    rank whose WORLD_SIZE disagrees with ``--gpus`` exits non-zero (the JSON's n_gpus must equal what was asked for)."""
    rank, local_rank, world = rank_info()
    if "WORLD_SIZE" not in os.environ and int(gpus_arg) > 1:
        raise SystemExit(self_launch(int(gpus_arg), script, argv))
    if world != int(gpus_arg):
        sys.stderr.write(f"{os.path.basename(script)}: --gpus {gpus_arg} but WORLD_SIZE={world}\n")
        raise SystemExit(2)
    return rank, local_rank, world


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
