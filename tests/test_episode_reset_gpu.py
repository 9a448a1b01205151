"""Episode reset inside a batch: BaseEnv.reset (env.py:30-51) -> PhysTwinDynamics.reset (phystwin.py:39-102: a NEW dynamics module,
i.e. particles in their start pose at rest, current_openness = None, grasped = False, zero collision forces) + GSRenderer.reset_state,
for SOME environments of a BatchedRollout while the others keep running (episodes of eval_policy_parallel.py:266-280 are independent
and end at different steps)."""
import numpy as np
import pytest

from util_parity import record

pytestmark = pytest.mark.gpu


def _frac_differing(a, b, atol=1e-4):
    return float(((a - b).abs() > atol).float().mean().item())


def test_reset_of_one_environment_restarts_its_episode_and_leaves_the_others_alone():
    import torch
    from r2s_hip.rollout import BatchedRollout

    kw = dict(num_substeps=20, seed=11, n_env=3)
    a, b = BatchedRollout("tiny", **kw), BatchedRollout("tiny", **kw)
    xs, cols = [], []
    for _ in range(6):                                     # the twin never resets: what every episode looks like, step by step
        b.step()
        col, _ = b.observations()
        torch.cuda.synchronize()
        xs.append(b.phys.x.clone()); cols.append(col.clone())
    x_init, v_init = a.phys.x.clone(), a.phys.v.clone()
    for _ in range(3):
        a.step()
    torch.cuda.synchronize()
    assert torch.equal(a.phys.x, xs[2]), "same seed, same steps: the twins agree bit for bit before the reset"
    assert (a.phys.x[1] - x_init[1]).abs().max() > 1e-5, "the episode must have moved the rope"
    a.reset([1])
    torch.cuda.synchronize()
    assert torch.equal(a.phys.x[1], x_init[1]) and torch.equal(a.phys.v[1], v_init[1])
    assert torch.equal(a.phys.x[0], xs[2][0]) and torch.equal(a.phys.x[2], xs[2][2]), "the other environments keep their state"
    assert torch.equal(a.eef_xyz[1], b._init["eef_xyz"][1]) and not torch.equal(a.eef_xyz[0], b._init["eef_xyz"][0])
    opn, grasped = a.phys.eef_state()
    assert float(opn[1]) == 0.0 and float(opn[0]) > 0.0 and float(opn[2]) > 0.0 and grasped.tolist() == [0, 0, 0], (opn, grasped)   # current_openness: None again for env 1 only
    assert float(a.phys.collision_forces()[1].abs().max()) == 0.0
    worst_x, worst_px = 0.0, 0.0
    for k in range(3):
        a.step()
        col, _ = a.observations()
        torch.cuda.synchronize()
        # env 1 lives its first steps again (same start state, same trace from its start), envs 0 and 2 their steps 4..6
        for e, ref in ((1, k), (0, 3 + k), (2, 3 + k)):
            dx = float((a.phys.x[e] - xs[ref][e]).abs().max())
            worst_x = max(worst_x, dx)
            assert dx <= 2e-6, (k, e, dx)
            fr = _frac_differing(col[e], cols[ref][e])
            worst_px = max(worst_px, fr)
            assert fr <= 1e-3, (k, e, fr)
    assert a.lossy_batches == 0
    record("episode reset of one environment of three", x_max_abs_vs_unreset_twin=worst_x, differing_pixel_fraction=worst_px, tol=2e-6)


def test_reset_of_all_environments_while_a_candidate_rebuild_is_in_flight_and_in_pipelined_mode():
    """reset() right after step(): the candidate rebuild of that step is still running on the side stream and (pipelined mode) the
    render stream still reads the Gaussians — the reset must order itself behind both.  Two episodes of the same batch are equal."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("tiny", num_substeps=20, seed=4, n_env=2)
    ro.set_pipelined(True)

    def episode():
        out = []
        for _ in range(4):
            ro.step()
            col, _ = ro.observations()
            out.append((ro.phys.x.clone(), col.clone()))
        return out

    first = episode()
    ro.reset()                       # no synchronisation in between
    second = episode()
    torch.cuda.synchronize()
    for k, ((x1, c1), (x2, c2)) in enumerate(zip(first, second)):
        assert float((x1 - x2).abs().max()) <= 2e-6, k
        assert _frac_differing(c1, c2) <= 1e-3, k
    assert bool(torch.isfinite(ro.phys.x).all()) and np.isfinite(float(second[-1][1].sum()))


def test_get_obs_mirrors_base_env_get_obs_for_the_batch():
    """BaseEnv.get_obs (env.py:53-74): fixed-camera and wrist-camera image lists + obs['robot'] = the state an action left behind
    (PhysTwinDynamics.step's next_state, phystwin.py:158-167: eef_xyz_next, quaternion of eef_rot_next, the commanded opening)."""
    import torch
    from scipy.spatial.transform import Rotation
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("tiny", num_substeps=10, seed=2, n_env=2)
    rv = np.array([[0.0, 0.02, 0.3], [0.25, -0.1, 0.0]])
    rot0 = ro.eef_rot.cpu().numpy().astype(np.float64)
    rot_next = np.stack([Rotation.from_rotvec(rv[e]).as_matrix().T @ rot0[e] for e in range(2)])
    xyz_next = ro.eef_xyz + torch.tensor([[0.001, 0.0, -0.002], [0.0, 0.0015, 0.001]], device=ro.device)
    act = torch.cat([xyz_next, torch.from_numpy(rot_next).float().to(ro.device).reshape(2, 9), torch.tensor([[0.7], [0.4]], device=ro.device)], 1)
    ro.step(act)
    obs = ro.get_obs()
    torch.cuda.synchronize()
    assert len(obs["image_list"]) == 1 and len(obs["image_wrist_list"]) == 1 and obs["image_list"][0].shape == (2, 3, ro.H, ro.W)
    assert obs["depth_wrist_list"][0].shape == (2, 1, ro.H, ro.W)
    assert torch.equal(obs["image_list"][0], ro.out_color[:, 0]) and torch.equal(obs["image_wrist_list"][0], ro.out_color[:, 1])
    assert float(obs["image_list"][0].std()) > 0
    r = obs["robot"]
    assert torch.equal(r["eef_xyz"], xyz_next) and torch.equal(r["eef_gripper"], act[:, 12:13])
    q = r["eef_quat"].cpu().numpy().astype(np.float64)
    ref = Rotation.from_matrix(rot_next).as_quat()[:, [3, 0, 1, 2]]
    assert np.abs(q - np.sign((q * ref).sum(1, keepdims=True)) * ref).max() < 2e-6
    assert (q[:, 0] > 0).all()                     # small rotations: the trace branch, scalar part positive


def test_episode_scheduler_on_the_real_rollout():
    """r2s_hip.evaluate.run_episodes (the loop of experiments/eval_policy_parallel.py:40-255 over the slots of a batch) on the HIP
    rollout: five episodes on two environments = three waves of resets; a 13-dim policy action per step, the settling steps hold
    the pose (the end effector does not move during them), every episode of a slot starts from the same state."""
    import torch
    from r2s_hip.evaluate import hold_pose_action, run_episodes, summarize
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("tiny", num_substeps=10, seed=6, n_env=2)
    start_xyz = ro._init["eef_xyz"].clone()
    seen = []

    def policy(obs, episode_step, active):
        a = hold_pose_action(obs)
        a[:, 2] -= 0.001                       # 1 mm down per step
        a[:, 12] = 0.8
        return a

    def on_step(r, slot_episode, episode_step):
        seen.append((slot_episode.tolist(), episode_step.tolist(), r.eef_xyz.clone(), r.phys.x.clone()))

    rec = run_episodes(ro, [3, 4, 5, 6, 7], policy=policy, max_steps=3, settle_steps=2, on_step=on_step)
    torch.cuda.synchronize()
    assert rec[:, 0].tolist() == [3, 4, 5, 6, 7] and rec[:, 2].tolist() == [3.0] * 5 and (rec[:, 3] > 0).all()
    assert len(seen) == 3 * (2 + 3)
    assert [s[0] for s in seen[::5]] == [[3, 4], [5, 6], [7, -1]]
    for w in range(3):
        first = seen[5 * w]
        assert torch.allclose(first[2][0], start_xyz[0], atol=1e-7), "the settling steps hold the start pose"
        last = seen[5 * w + 4]
        assert abs(float(last[2][0, 2] - start_xyz[0, 2]) + 0.003) < 1e-6       # three policy steps of 1 mm
        if w:
            assert torch.equal(first[3][0], seen[0][3][0]), "every episode of a slot starts from the same state"
    assert bool(torch.isfinite(ro.phys.x).all()) and summarize(rec)["episodes"] == 5
