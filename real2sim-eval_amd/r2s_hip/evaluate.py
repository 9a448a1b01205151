"""experiments/eval_policy_parallel.py for batches: episodes shard over the ranks exactly as the reference deals them to its worker
processes (episode e -> rank e % n_processes, :266-271), and inside a rank the episodes are dealt to the environment slots of one
``BatchedRollout`` instead of being run one after the other (:40-255: reset -> 30 holding steps -> policy loop until the step
limit).  No data-path collective: ranks meet once, at the end, to all-gather one fixed-size record per episode — {episode_id,
success, steps, wall_ms} — so that every rank holds the global success rate.

The rollout is duck-typed (``n_env``, ``reset(mask)`` — ``reset(mask, episode_ids=[...])`` when it has a true ``randomize`` attribute:
the episode id picks the object's start pose like the reference's ``env.reset(seed=episode_id)`` —, ``get_obs()``, ``step(action)``,
``success_flags()``), so the scheduler is
covered on the CPU with a stand-in (tests/test_evaluate.py) and on the GPU with the real one (tests/test_episode_reset_gpu.py)."""
from __future__ import annotations

import time
from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist

RECORD_FIELDS = ("episode_id", "success", "steps", "wall_ms")


def episodes_of_rank(n_episodes: int, rank: int, world: int):
    """eval_policy_parallel.py:266-271: ``episodes % n_processes == rank``."""
    return [e for e in range(int(n_episodes)) if e % int(world) == int(rank)]


def hold_pose_action(obs) -> torch.Tensor:
    """The action the reference holds for its 30 stabilising steps after a reset (:88-108): the current pose, [n_env, 13]."""
    from .rollout import quaternion_to_rotation_matrix

    r = obs["robot"]
    rot = quaternion_to_rotation_matrix(r["eef_quat"])
    return torch.cat([r["eef_xyz"], rot.reshape(rot.shape[0], 9), r["eef_gripper"].reshape(-1, 1)], 1)


def run_episodes(ro, episode_ids: Sequence[int], policy: Optional[Callable] = None, max_steps: int = 100, settle_steps: int = 30,
                 stop_on_success: bool = False, on_step: Optional[Callable] = None) -> torch.Tensor:
    """Run ``episode_ids`` on the ``ro.n_env`` environment slots of one rank.  Returns float64 records [len(episode_ids), 4]
    (RECORD_FIELDS), in the order of ``episode_ids``.

    A slot's life: take the next episode -> ``ro.reset(slot)`` -> ``settle_steps`` steps holding the pose (not counted; :106-108) ->
    policy steps; the episode ends at ``max_steps`` (the reference's only end: the TimeLimit wrapper's `truncated`, env.py:12) or,
    with ``stop_on_success``, at the first step whose state satisfies the task predicate.  ``success`` is the predicate on the
    episode's last state.  ``policy(obs, episode_step [n_env] long, active [n_env] bool) -> action``: an [n_env, 13] 'xyz_rot'
    tensor (rows of settling or idle slots are replaced by "hold the current pose"), a motion dict (passed through), or None = the
    rollout's own synthetic trace, which restarts with every reset (also what ``policy=None`` means).  Slots without an episode
    (the tail of the list) are not recorded.  ``on_step(ro, slot_episode [n_env] long, episode_step)`` is the hook for an
    observation sink.

    Host synchronisation: none of its own while ``stop_on_success`` is off (every end is decided by host-side counters; the success
    flags of ending episodes are collected on the device and read by whoever reads the records) — ``get_obs`` waits for the frame it
    hands out, as a closed loop must; one [n_env] read per step with ``stop_on_success``."""
    E = int(ro.n_env)
    ids = [int(e) for e in episode_ids]
    n = len(ids)
    dev = ro.success_flags().device
    rec = torch.zeros(n, 4, dtype=torch.float64, device=dev)
    if n:
        rec[:, 0] = torch.tensor(ids, dtype=torch.float64, device=dev)
    slot_row = [-1] * E                 # index into `ids` of the episode a slot runs (-1: idle)
    slot_step = [0] * E                 # policy steps taken by that episode (negative: still settling)
    slot_t0 = [0.0] * E
    next_row = 0

    def deal(slots):
        nonlocal next_row
        mask = torch.zeros(E, dtype=torch.bool)
        for s in slots:
            if next_row < n:
                slot_row[s], slot_step[s], slot_t0[s] = next_row, -int(settle_steps), time.perf_counter()
                next_row += 1
                mask[s] = True
            else:
                slot_row[s] = -1
        if bool(mask.any()):
            if getattr(ro, "randomize", False):
                # env.reset(seed=episode_id) (eval_policy.py:67, eval_policy_parallel.py:47): the episode id is the index of the object's randomised start pose
                ro.reset(mask, episode_ids=[ids[slot_row[s]] if slot_row[s] >= 0 else 0 for s in range(E)])
            else:
                ro.reset(mask)      # a HOST mask: a deal that covers every slot is a full reset (clears a sticky fault word)

    deal(range(E))
    while any(r >= 0 for r in slot_row):
        obs = ro.get_obs()
        act_host = [r >= 0 and st >= 0 for r, st in zip(slot_row, slot_step)]
        active = torch.tensor(act_host, dtype=torch.bool, device=dev)
        step_t = torch.tensor([max(st, 0) for st in slot_step], dtype=torch.long, device=dev)
        action = policy(obs, step_t, active) if policy is not None else None
        if torch.is_tensor(action) and obs.get("robot") is not None and not all(act_host):
            # slots that are settling after a reset (or idle) hold their pose, whatever the policy said about them
            action = torch.where(active[:, None], action.to(dev, torch.float32).reshape(E, 13), hold_pose_action(obs))
        ro.step(action)
        if on_step is not None:
            on_step(ro, torch.tensor([ids[r] if r >= 0 else -1 for r in slot_row], dtype=torch.long), step_t)
        for s in range(E):
            if slot_row[s] >= 0:
                slot_step[s] += 1
        flags = None
        ended = [s for s in range(E) if slot_row[s] >= 0 and slot_step[s] >= int(max_steps)]
        if stop_on_success:
            flags = ro.success_flags()
            host = flags.cpu().tolist()                                  # the one read per step of this mode
            ended = sorted(set(ended) | {s for s in range(E) if slot_row[s] >= 0 and slot_step[s] > 0 and host[s]})
        if ended:
            # Every running episode ends with this step: what follows is a FULL reset (which hands the stepper a whole new state and with it
            # clears its sticky fault word) or the end of the run — and the stepper reports a fault with a lag of up to two env steps.  A
            # fault raised in the last steps of these episodes (a hand-off that timed out, an impulse beyond the bound a skipped mesh test
            # relies on: the state is invalid) would never surface and their outcome would be recorded from an invalid state.  Ask now
            # (one host synchronisation per wave of episodes; the rollout raises): nothing is recorded from such a state.
            if len(ended) == sum(r >= 0 for r in slot_row) and hasattr(ro, "check_fault"):
                ro.check_fault()
            if flags is None:
                flags = ro.success_flags()
            now = time.perf_counter()
            rows = torch.tensor([slot_row[s] for s in ended], dtype=torch.long, device=dev)
            sl = torch.tensor(ended, dtype=torch.long, device=dev)
            rec[rows, 1] = flags[sl].to(torch.float64)
            rec[rows, 2] = torch.tensor([float(slot_step[s]) for s in ended], dtype=torch.float64, device=dev)
            rec[rows, 3] = torch.tensor([(now - slot_t0[s]) * 1e3 for s in ended], dtype=torch.float64, device=dev)
            deal(ended)
    return rec


def gather_episode_records(records: torch.Tensor, n_episodes: int, device=None) -> torch.Tensor:
    """The one collective of an evaluation: every rank contributes its episodes' records, padded to the largest
    share, and receives the table of all ``n_episodes`` sorted by episode id — [n_episodes, 4] float64 on ``device``."""
    device = records.device if device is None else torch.device(device)
    records = records.to(device=device, dtype=torch.float64).reshape(-1, 4)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = records
    else:
        world = dist.get_world_size()
        share = -(-int(n_episodes) // world)
        pad = torch.full((share, 4), -1.0, dtype=torch.float64, device=device)
        pad[: records.shape[0]] = records
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out = torch.cat(parts)
        out = out[out[:, 0] >= 0]
    order = torch.argsort(out[:, 0])
    return out[order]


def summarize(table: torch.Tensor) -> dict:
    t = table.cpu()
    n = int(t.shape[0])
    return dict(episodes=n, success_rate=float(t[:, 1].mean().item()) if n else 0.0, mean_steps=float(t[:, 2].mean().item()) if n else 0.0,
                mean_wall_ms=float(t[:, 3].mean().item()) if n else 0.0)
