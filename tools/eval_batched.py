#!/usr/bin/env python
"""experiments/eval_policy_parallel.py for batches: N ranks (one per GPU), each with one BatchedRollout; the episodes are dealt to the
ranks as the reference deals them to its worker processes (episode e -> rank e % N, :266-271), inside a rank to the environment slots
of its batch (r2s_hip.evaluate.run_episodes: per-slot reset -> holding steps -> policy steps -> record), and the ranks meet ONCE, to
all-gather the per-episode records {episode_id, success, steps, wall_ms} (RCCL on GPUs).  Rank 0 prints one JSON line.

    python tools/eval_batched.py --config T_pusher_32env --episodes 256 --max-steps 100 --gpus 8
    python tools/eval_batched.py --stub --episodes 13 --gpus 2          # CPU stand-in over gloo (the launcher / scheduler / gather path)

The policy here is a scripted one — the rollout's own synthetic action trace (which restarts with every episode) — because policy
inference is outside the scope of this repository (out of scope); a real policy is a callable
``policy(obs, episode_step, active) -> [n_env, 13]`` handed to ``r2s_hip.evaluate.run_episodes``."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "real2sim-eval_amd"), ROOT]


class StubRollout:
    """CPU stand-in with the rollout's episode interface: environment s 'succeeds' after 3 + s % 2 steps of an episode."""

    def __init__(self, n_env):
        import torch

        self.n_env = n_env
        self.age = torch.zeros(n_env, dtype=torch.long)

    def reset(self, mask):
        self.age[mask] = 0

    def get_obs(self):
        return dict(image_list=[], image_wrist_list=[], robot=None)

    def step(self, action=None):
        time.sleep(0.001)
        self.age += 1

    def success_flags(self):
        import torch

        return self.age >= 3 + (torch.arange(self.n_env) % 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", default="T_pusher_32env")
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default: the config's)")
    ap.add_argument("--episodes", type=int, default=64)
    ap.add_argument("--max-steps", type=int, default=60)
    ap.add_argument("--settle-steps", type=int, default=0, help="holding steps after a reset (the reference: 30; the synthetic scenes start at rest)")
    ap.add_argument("--substeps", type=int, default=667)
    ap.add_argument("--stop-on-success", action="store_true")
    ap.add_argument("--no-randomize", action="store_true", help="every episode from the same start pose (default: the episode id indexes the object's "
                                                                "grid pose like env.reset(seed=episode_id), eval_policy_parallel.py:47)")
    ap.add_argument("--stub", action="store_true", help="CPU stand-in rollout over gloo")
    args = ap.parse_args()

    from r2s_hip import dist as rdist
    from r2s_hip import evaluate as ev

    rank, local_rank, world = rdist.resolve_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    import torch
    import torch.distributed as dist

    if args.stub:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        ro = StubRollout(args.envs or 4)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("eval_batched.py needs an MI355X (or --stub): there is no CPU fallback for the product path")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        from r2s_hip.rollout import BatchedRollout

        ro = BatchedRollout(args.config, device=dev, seed=rank, n_env=args.envs, num_substeps=args.substeps, randomize=not args.no_randomize)
    mine = ev.episodes_of_rank(args.episodes, rank, world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    rec = ev.run_episodes(ro, mine, policy=None, max_steps=args.max_steps, settle_steps=args.settle_steps, stop_on_success=args.stop_on_success)
    if not args.stub:
        torch.cuda.synchronize(dev)
    elapsed = rdist.max_over_ranks(time.perf_counter() - t0, dev)
    table = ev.gather_episode_records(rec, args.episodes, dev)                 # the one collective
    if rank == 0:
        s = ev.summarize(table)
        env_steps = float(table[:, 2].sum().item()) + args.settle_steps * int(table.shape[0])
        print(json.dumps({"workload": "stub" if args.stub else args.config, "n_gpus": world, "envs_per_gpu": ro.n_env, **s,
                          "episodes_per_rank": [len(ev.episodes_of_rank(args.episodes, r, world)) for r in range(world)],
                          "elapsed_s": elapsed, "env_steps_per_s": env_steps / elapsed if elapsed > 0 else 0.0,
                          "episode_ids_seen": int(table.shape[0]),
                          "randomized": bool(getattr(ro, "randomize", False)),   # object start pose = grid pose of the episode id (rank 0's episodes listed)
                          "random_variables": {str(k): v for k, v in sorted(getattr(ro, "random_variables", {}).items())[:16]},
                          "collective": "one all_gather of [ceil(episodes / ranks), 4] float64 per rank"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
