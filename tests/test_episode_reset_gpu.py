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


def test_get_obs_renders_the_start_state_and_the_state_a_reset_left():
    """the reference's env.reset() returns get_obs() rendered FROM the reset state (env.py:30-51).  The first
    get_obs() of a rollout must show the start state (not the uninitialised output arrays), and the first get_obs() after a partial
    reset must show the reset environment as a fresh rollout shows it — next to the other environments' current frames."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    kw = dict(num_substeps=20, seed=11, n_env=3)
    a, b = BatchedRollout("tiny", **kw), BatchedRollout("tiny", **kw)
    a.out_color.fill_(float("nan"))                       # whatever torch.empty held: must never reach a caller
    first_a = [t.clone() for t in a.get_obs()["image_list"] + a.get_obs()["image_wrist_list"]]
    first_b = [t.clone() for t in b.get_obs()["image_list"] + b.get_obs()["image_wrist_list"]]
    torch.cuda.synchronize()
    for ia, ib in zip(first_a, first_b):
        assert bool(torch.isfinite(ia).all()) and float(ia.std()) > 0
        assert torch.equal(ia, ib), "same seed: the two start frames are the same frame"
    for _ in range(3):
        a.step()
    last = a.get_obs()
    last_side, last_wrist = last["image_list"][0].clone(), last["image_wrist_list"][0].clone()
    assert _frac_differing(last_wrist[1], first_a[1][1]) > 1e-3, "three steps must have changed environment 1's wrist view"
    a.reset([1])
    obs = a.get_obs()                                      # no step in between
    torch.cuda.synchronize()
    side, wrist = obs["image_list"][0], obs["image_wrist_list"][0]
    assert _frac_differing(side[1], first_b[0][1]) <= 1e-3 and _frac_differing(wrist[1], first_b[1][1]) <= 1e-3, "reset slot: the start frame"
    for e in (0, 2):
        assert _frac_differing(side[e], last_side[e]) <= 1e-3 and _frac_differing(wrist[e], last_wrist[e]) <= 1e-3, "running slots: their current frame"
    assert torch.equal(a.robot_state()["eef_xyz"][1], b._init["eef_xyz"][1])


def test_a_fault_of_a_running_environment_survives_the_reset_of_another(monkeypatch):
    """the sticky fault word is per handle.  An episode reset of environment 1 (r2s_phys_set_state_envs) must
    not clear a fault raised by environment 0, which keeps stepping on invalid state otherwise; a whole new state does clear it."""
    import torch
    from r2s_hip import synth
    from r2s_hip._lib import R2SError
    from util_physics import far_apart, gripper_motion, hip_env, two_sheets

    monkeypatch.setenv("R2S_MESH_DEFER", "1")
    n_sub = 4
    ob, nA = two_sheets()
    far = far_apart(ob, nA)
    c = ob["points"].mean(0)
    fingers = [synth.finger_mesh((c[0], c[1] - 0.05, c[2])), synth.finger_mesh((c[0], c[1] + 0.05, c[2]))]
    interp, centers, dv, om = gripper_motion(fingers, n_sub, 5e-5, vel=(0.0, 0.0, 0.0), closing=0.0)
    h = hip_env(far, n_env=2, num_substeps=n_sub, dynamic_meshes=fingers, self_collision=True)
    x0 = torch.from_numpy(ob["points"])[None].repeat(2, 1, 1)
    v0 = torch.zeros_like(x0)
    toward = np.sign(ob["points"][nA:, 1].mean() - ob["points"][:nA, 1].mean())
    v0[0, :nA, 1] = 200.0 * float(toward)                  # environment 0 only: impulses beyond the 40 m/s bound
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].repeat(2, *([1] * a.ndim)).cuda()  # noqa: E731
    h.set_state(x0, v0)
    h.update_collision_graph()
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    h.step()
    torch.cuda.synchronize()
    h.set_state_envs(x0, torch.zeros_like(x0), torch.tensor([0, 1]))       # episode reset of environment 1
    torch.cuda.synchronize()
    with pytest.raises(R2SError, match="40 m/s"):
        h.step()
    assert torch.equal(h.x[1].cpu(), x0[1]) and float(h.v[1].abs().max()) == 0.0
    h.set_state(x0, torch.zeros_like(x0))                   # every environment: usable again
    h.update_collision_graph()
    h.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(h.x).all())


def test_episode_ids_place_the_object_at_the_grid_pose_of_the_episode():
    """env.reset(seed=episode_id) -> load_scaniverse(randomize, index) (env.py:30-34, gs_renderer.py:340-347,
    :614-637): the episode id indexes the object's grid pose.  Episodes of one batch must start from DIFFERENT poses, an episode's
    pose must not depend on the slot it runs in, and random_variables must record [x, y, z, angle]."""
    import torch
    from r2s_hip.evaluate import run_episodes
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("tiny", num_substeps=10, seed=6, n_env=2, randomize=True)
    base = ro._init["x"][0].clone()
    c = ro._obj_center
    seen = {}

    def on_step(r, slot_episode, episode_step):
        for s, ep in enumerate(slot_episode.tolist()):
            if ep >= 0 and ep not in seen:
                seen[ep] = (s, r.bones[s].clone(), r.means[s, : r.n_obj].clone(), r.eef_xyz[s].clone())

    ids = [0, 4, 13, 27, 31]
    poses = {e: ro.episode_pose(e) for e in ids}
    assert poses[0] == (-0.05, -0.05, 0.0, -10 * np.pi / 180) and poses[4][:2] == (-0.05, 0.0) and abs(poses[4][3]) < 1e-12      # rope grid: 9 xy x 3 theta
    assert poses[27] == poses[0] and poses[13] != poses[0]
    # reset by hand for exact checks (run_episodes below exercises the scheduler path)
    ro.reset([0, 1], episode_ids=[13, 4])
    torch.cuda.synchronize()
    for s, ep in ((0, 13), (1, 4)):
        x, y, _, a = poses[ep]
        Rz = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32, device=ro.device)
        want = (base - c) @ Rz.T + c + torch.tensor([x, y, 0.0], device=ro.device)
        assert float((ro.phys.x[s] - want).abs().max()) < 2e-6
        assert float((ro.bones[s] - want).abs().max()) < 2e-6
        assert ro.random_variables[ep] == [float(np.float32(v)) for v in poses[ep]]
    assert float((ro.phys.x[0] - ro.phys.x[1]).abs().max()) > 0.02, "two episodes, two start poses"
    col = ro.get_obs()["image_list"][0]
    assert _frac_differing(col[0], col[1]) > 1e-3
    rec = run_episodes(ro, ids, policy=None, max_steps=2, settle_steps=0, on_step=on_step)
    torch.cuda.synchronize()
    assert rec[:, 0].tolist() == ids and set(ro.random_variables) >= set(ids)
    assert bool(torch.isfinite(ro.phys.x).all())
    # the same episode in another slot of another rollout starts from the same pose
    ro2 = BatchedRollout("tiny", num_substeps=10, seed=6, n_env=2, randomize=True)
    ro2.reset([1], episode_ids=[0, 13])
    torch.cuda.synchronize()
    ro.reset([0], episode_ids=[13, 0])
    torch.cuda.synchronize()
    assert float((ro2.phys.x[1] - ro.phys.x[0]).abs().max()) < 2e-6


def test_episode_index_above_the_object_grid_reposes_the_box_obstacle_and_the_rod_starts_clear_of_the_turned_block():
    """load_scaniverse peels the episode index — object pose = index mod n_object_rand, then mesh by mesh from index //
    n_object_rand (gs_renderer.py:340-383): the sloth scene has 5 object poses x 4 box poses = 20 start scenes, and episodes 0 and 5 differ
    (in the box).  A posed reset re-poses the static box of the stepper (r2s_phys_set_static_mesh_points) and the success predicate's box;
    the pusher's rod starts in front of the TURNED block, never inside it."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("sloth_32env", n_env=2, num_substeps=20, seed=3, randomize=True, settle_steps=0)
    # the index arithmetic, against the reference's loops restated literally
    n_obj, box = 5, BatchedRollout.GRIDS["sloth"]["meshes"][0]
    for idx in range(0, 23):
        true_index, true_index_mesh = idx % n_obj, idx // n_obj
        this = true_index_mesh % len(box["xy"])
        assert ro.episode_pose(idx)[:2] == tuple(float(v) for v in BatchedRollout.GRIDS["sloth"]["xy"][true_index])
        (mx, my, mz, ma), = ro.episode_mesh_poses(idx)
        assert (mx, my) == tuple(float(v) for v in box["xy"][this]) and abs(ma - box["theta"][this] * np.pi / 180) < 1e-12
    assert ro.episode_pose(0) == ro.episode_pose(5) and ro.episode_mesh_poses(0) != ro.episode_mesh_poses(5)
    assert ro.episode_mesh_poses(20) == ro.episode_mesh_poses(0)
    c0 = ro._box_c.clone()
    ro.reset([0, 1], episode_ids=[5, 12])            # box poses 1 (-5 cm in x, -5 deg) and 2 (+5 cm, +5 deg)
    torch.cuda.synchronize()
    assert torch.allclose(ro._box_c[0] - c0[0], torch.tensor([-0.05, 0.0, 0.0], device=ro.device), atol=1e-7)
    assert torch.allclose(ro._box_c[1] - c0[1], torch.tensor([0.05, 0.0, 0.0], device=ro.device), atol=1e-7)
    assert ro.random_mesh_variables[5][0][:2] == [-0.05, 0.0] and ro.random_mesh_variables[12][0][:2] == [0.05, 0.0]
    # the predicate follows the box: particles gathered inside environment 0's RE-POSED box satisfy it, inside the old pose they do not
    half = torch.tensor(ro._box[1], dtype=torch.float32, device=ro.device)
    x = ro.phys.x.clone()
    x[0] = ro._box_c[0] + (torch.rand(ro.N, 3, device=ro.device) - 0.5) * half
    x[1] = c0[1] + torch.tensor([-0.12, 0.0, 0.0], device=ro.device) + (torch.rand(ro.N, 3, device=ro.device) - 0.5) * 0.01   # 7 cm outside the box moved to +5 cm
    ro.phys.set_state(x)
    assert ro.success_flags().tolist() == [True, False]
    for _ in range(2):
        ro.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ro.phys.x).all())
    # pusher: the rod's start pose is derived from the turned block's extent
    rp = BatchedRollout("T_pusher_32env", n_env=9, num_substeps=20, seed=3, randomize=True, close_at=3)
    rp.reset(list(range(9)), episode_ids=list(range(9)))       # 4 xy x 4 theta: angles 45 .. 315 degrees
    torch.cuda.synchronize()
    x = rp.phys.x
    gap = x[:, :, 0].min(1).values - rp.eef_xyz[:, 0]
    assert float(gap.min()) > 0.005 + 0.001, ("the rod (radius 5 mm, margin 1 mm) must start clear of the block's -x extent", gap.tolist())
    assert float((gap - gap[0]).abs().max()) < 1e-5, "the same clearance for every pose"
    rp.step(); rp.get_obs()
    assert rp.contact_stats()["mesh_contacts"] == 0, "no penetration at the start of a posed episode"
