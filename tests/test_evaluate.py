"""The episode scheduler of r2s_hip/evaluate.py (experiments/eval_policy_parallel.py for batches: episodes e -> rank e % world,
:266-271; per episode reset -> holding steps -> policy loop -> record; ONE all-gather of the records at the end) on
the CPU with a stand-in rollout; the real rollout runs the same scheduler in tests/test_episode_reset_gpu.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "real2sim-eval_amd"))


class FakeRollout:
    """n_env independent 1-D 'robots': the action's x is where the end effector goes; success = x >= 0.35."""

    def __init__(self, n_env):
        self.n_env = n_env
        self.x = torch.full((n_env,), 9.0)            # garbage until the first reset
        self.log = []

    def reset(self, mask):
        assert mask.dtype == torch.bool and mask.shape == (self.n_env,)
        self.x = torch.where(mask, torch.zeros_like(self.x), self.x)
        self.log.append(("reset", mask.tolist()))

    def get_obs(self):
        E = self.n_env
        xyz = torch.stack([self.x, torch.zeros(E), torch.zeros(E)], 1)
        quat = torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(E, 1)
        return dict(image_list=[], image_wrist_list=[], robot=dict(eef_xyz=xyz, eef_quat=quat, eef_gripper=torch.ones(E, 1)))

    def step(self, action):
        assert action.shape == (self.n_env, 13)
        assert torch.equal(action[:, 3:12], torch.eye(3).reshape(1, 9).repeat(self.n_env, 1))
        self.log.append(("step", action[:, 0].tolist()))
        self.x = action[:, 0].clone()

    def success_flags(self):
        return self.x >= 0.35


def _policy(obs, episode_step, active):
    """Slot s advances by 0.1 (even slots) or 0.2 (odd slots) per step — and asks idle / settling slots to jump to 5.0, which
    the scheduler must override with 'hold the pose'."""
    from r2s_hip.evaluate import hold_pose_action

    a = hold_pose_action(obs)
    E = a.shape[0]
    speed = torch.tensor([0.1 if s % 2 == 0 else 0.2 for s in range(E)])
    a[:, 0] = torch.where(active, a[:, 0] + speed, torch.full((E,), 5.0))
    return a


def test_episodes_are_dealt_like_the_reference_deals_them_to_its_workers():
    from r2s_hip.evaluate import episodes_of_rank

    assert episodes_of_rank(7, 0, 2) == [0, 2, 4, 6] and episodes_of_rank(7, 1, 2) == [1, 3, 5]
    assert sorted(sum((episodes_of_rank(256, r, 8) for r in range(8)), [])) == list(range(256))
    assert len(episodes_of_rank(256, 3, 8)) == 32 and episodes_of_rank(3, 5, 8) == []


def test_scheduler_fixed_length_episodes_like_the_reference():
    """The reference's only episode end is the step limit: every episode runs settle + max_steps steps, success is read from the
    last state, slots restart together; five episodes on two slots = three waves, the last with one idle slot."""
    from r2s_hip.evaluate import run_episodes, summarize

    ro = FakeRollout(2)
    rec = run_episodes(ro, [10, 11, 12, 13, 14], policy=_policy, max_steps=3, settle_steps=2)
    assert rec[:, 0].tolist() == [10, 11, 12, 13, 14]
    assert rec[:, 2].tolist() == [3.0] * 5
    # slot 0 (episodes 10, 12, 14) reaches 0.3: no success; slot 1 (11, 13) reaches 0.6
    assert rec[:, 1].tolist() == [0.0, 1.0, 0.0, 1.0, 0.0]
    assert (rec[:, 3] > 0).all()
    resets = [m for k, m in ro.log if k == "reset"]
    assert resets == [[True, True], [True, True], [True, False]]
    steps = [x for k, x in ro.log if k == "step"]
    assert len(steps) == 3 * (2 + 3)
    assert steps[0] == [0.0, 0.0] and steps[1] == [0.0, 0.0]          # settling: the pose is held although the policy said 5.0
    assert steps[2] == pytest.approx([0.1, 0.2])
    assert steps[-1][1] == pytest.approx(0.6)                          # the idle slot of the last wave holds where its last episode ended
    s = summarize(rec)
    assert s["episodes"] == 5 and s["success_rate"] == pytest.approx(0.4) and s["mean_steps"] == 3.0


def test_scheduler_stop_on_success_staggers_the_slots():
    from r2s_hip.evaluate import run_episodes

    ro = FakeRollout(2)
    rec = run_episodes(ro, list(range(6)), policy=_policy, max_steps=10, settle_steps=1, stop_on_success=True)
    assert rec[:, 0].tolist() == [0, 1, 2, 3, 4, 5]
    assert rec[:, 1].tolist() == [1.0] * 6
    # speed 0.1 needs 4 steps to pass 0.35, speed 0.2 needs 2: the fast slot takes more of the episodes
    by_steps = sorted(rec[:, 2].tolist())
    assert set(by_steps) == {2.0, 4.0} and by_steps.count(2.0) > by_steps.count(4.0)
    resets = [m for k, m in ro.log if k == "reset"]
    assert resets[0] == [True, True] and [False, True] in resets and sum(sum(m) for m in resets) == 6


def test_no_policy_means_the_rollouts_own_trace():
    from r2s_hip.evaluate import run_episodes

    class Traced(FakeRollout):
        def step(self, action):
            assert action is None
            self.x = self.x + 0.2

    ro = Traced(3)
    rec = run_episodes(ro, [0, 1, 2, 3], policy=None, max_steps=2, settle_steps=0)
    assert rec[:, 2].tolist() == [2.0] * 4 and rec[:, 1].tolist() == [1.0] * 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "real2sim-eval_amd"))
    from r2s_hip import evaluate as ev

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = ev.episodes_of_rank(7, rank, world)
    rec = ev.run_episodes(FakeRollout(2), mine, policy=_policy, max_steps=3, settle_steps=1)
    table = ev.gather_episode_records(rec, 7, "cpu")                  # the one collective
    q.put((rank, mine, table[:, :3].tolist(), ev.summarize(table)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_gather_one_table_of_all_episodes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, mine0, t0, s0), (r1, mine1, t1, s1) = res
    assert mine0 == [0, 2, 4, 6] and mine1 == [1, 3, 5]
    assert t0 == t1 and [row[0] for row in t0] == [0, 1, 2, 3, 4, 5, 6] and all(row[2] == 3.0 for row in t0)
    # on each rank the episodes alternate between its slot 0 (0.3: fail) and slot 1 (0.6: success)
    assert [row[1] for row in t0] == [0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0]
    assert s0 == s1 and s0["episodes"] == 7


@pytest.mark.timeout(300)
def test_eval_batched_cli_two_ranks_stub():
    """tools/eval_batched.py (the eval_policy_parallel.py of this repository) from a plain process with --gpus 2: self-launch through
    torch.distributed.run, episodes e -> rank e % 2, the scheduler on a stand-in rollout, ONE all-gather, one JSON line from rank 0."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_batched.py"), "--stub", "--episodes", "13", "--gpus", "2", "--envs", "3",
                        "--max-steps", "5", "--stop-on-success"], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["episodes"] == 13 and d["episode_ids_seen"] == 13 and d["episodes_per_rank"] == [7, 6]
    assert d["success_rate"] == 1.0 and 3.0 <= d["mean_steps"] <= 4.0 and d["env_steps_per_s"] > 0


def test_a_fault_pending_when_a_whole_wave_of_episodes_ends_is_raised_not_recorded():
    """the stepper reports a fault with a lag of two env steps and a full reset clears it — when every slot ends on
    the same step the scheduler must ask (check_fault) BEFORE it records the wave and resets everything."""
    from r2s_hip.evaluate import run_episodes

    class Faulty(FakeRollout):
        def __init__(self, n_env, fault_at_check):
            super().__init__(n_env)
            self.checks, self.fault_at_check = 0, fault_at_check

        def check_fault(self):
            self.checks += 1
            if self.checks == self.fault_at_check:
                raise RuntimeError("r2s_phys_check_fault: invalid argument a workgroup waited beyond the poll limit; the state is invalid")

    ok = Faulty(2, fault_at_check=0)
    rec = run_episodes(ok, [0, 1, 2, 3], policy=_policy, max_steps=3, settle_steps=1)
    assert ok.checks == 2 and rec[:, 2].tolist() == [3.0] * 4          # asked once per wave that ends together
    bad = Faulty(2, fault_at_check=2)
    with pytest.raises(RuntimeError, match="state is invalid"):
        run_episodes(bad, [0, 1, 2, 3], policy=_policy, max_steps=3, settle_steps=1)
    # slots that end at different steps: a partial reset keeps the fault word, the stepper's own lagged report is enough — no extra syncs
    stag = Faulty(2, fault_at_check=0)
    run_episodes(stag, [0, 1, 2], policy=_policy, max_steps=3, settle_steps=1, stop_on_success=True)
    assert stag.checks <= 2
