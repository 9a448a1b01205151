"""Row R0 on the device, per environment: the wrist camera that follows each environment's gripper (GSRenderer.render_wrist,
gs_renderer.py:953-1000 + setup_camera, transform_utils.py:7-31), the action-taking batched step (env.py:86-94 ->
phystwin.py:104-147), and the rasteriser reading the per-step matrices."""
import os

import numpy as np
import pytest

from test_wrist_camera_oracle import scaled_ulps
from util_parity import record
from util_raster import compare_images, oracle_render

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_wrist_camera_vs_the_reference_fixture():
    """tests/golden/wrist_camera.npz: 12 end-effector poses through the reference's own render_wrist + setup_camera.  The
    kernel evaluates the two inverses in float64 (the reference: float32 LAPACK / torch.inverse), so agreement is to a few
    float32 spacings of each array's largest entry; gate: 4."""
    import torch
    from r2s_hip.camera import WristCamera

    G = np.load(os.path.join(HERE, "golden", "wrist_camera.npz"))
    n = len(G["eef_xyz"])
    wc = WristCamera(n, int(G["W"]), int(G["H"]), G["K"], G["eef2c"], float(G["near"]), float(G["far"]))
    view, proj, pos = wc.update(torch.from_numpy(G["eef_xyz"]).cuda(), torch.from_numpy(G["eef_rot"]).cuda())
    torch.cuda.synchronize()
    view, proj, pos = view.cpu().numpy()[:, 0], proj.cpu().numpy()[:, 0], pos.cpu().numpy()
    assert wc.tanfovx == float(G["tanfovx"]) and wc.tanfovy == float(G["tanfovy"])
    worst = [0.0, 0.0, 0.0]
    for i in range(n):
        u = [scaled_ulps(view[i], G["viewmatrix"][i]), scaled_ulps(proj[i], G["projmatrix"][i]), scaled_ulps(pos[i], G["campos"][i])]
        worst = [max(a, b) for a, b in zip(worst, u)]
        assert max(u) <= 4, (i, u)
        assert np.array_equal(view[i][:, 3], np.array([0, 0, 0, 1], np.float32))      # the last row of w2c, exactly
    record("device wrist camera vs reference fixture", viewmatrix_scaled_ulps=worst[0], projmatrix_scaled_ulps=worst[1], campos_scaled_ulps=worst[2], tol=4)


def test_wrist_view_follows_each_environments_gripper_and_renders_like_the_oracle():
    import torch
    from oracle.camera_oracle import wrist_camera
    from r2s_hip import synth
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("tiny", num_substeps=20, seed=5, n_env=3)
    assert ro.wrist is not None and ro.views == 2
    seen = []
    for _ in range(3):
        ro.step()
        col, dep = ro.observations()
        torch.cuda.synchronize()
        seen.append(ro.wrist.viewmatrix.cpu().numpy().copy())
    assert np.abs(seen[0][0] - seen[0][1]).max() > 1e-3, "environments have their own wrist cameras"
    assert np.abs(seen[0][0] - seen[2][0]).max() > 1e-5, "the camera moves with the gripper"
    eef2c = np.linalg.inv(synth.WRIST_C2EEF) @ np.diag([1.0, -1.0, -1.0, 1.0])
    K = synth.scaled_K(synth.WRIST_K, ro.W, ro.H)
    xyz, rot = ro.eef_xyz.cpu().numpy(), ro.eef_rot.cpu().numpy()
    for e in range(ro.n_env):
        cam = ro.camera_numpy(e, 1)
        v, p, c = wrist_camera(xyz[e], rot[e], eef2c, K, ro.W, ro.H)
        assert scaled_ulps(cam["viewmatrix"][0], v) <= 4 and scaled_ulps(cam["projmatrix"][0], p) <= 4 and scaled_ulps(cam["campos"], c) <= 4
        _, col_ref, _, dep_ref = oracle_render(ro.scene_numpy(e), cam)
        r = compare_images(col[e, 1].cpu().numpy(), dep[e, 1].cpu().numpy(), col_ref, dep_ref, what=f"moving wrist camera, env {e}, vs oracle")
        assert r["frac_rgb"] <= 1e-3 and r["frac_depth"] <= 1e-3, (e, r)      # 160x120: one pixel is 5e-5 of the frame
        assert col_ref.std() > 0
    assert ro.lossy_batches == 0


def test_step_takes_actions_dict_and_xyz_rot_tensor():
    """BatchedRollout.step(action): (1) the synthetic trace handed in as an explicit per-environment action reproduces the
    default rollout bit for bit; (2) an [n_env, 13] 'xyz_rot' action (phystwin.py:113-118) moves every environment's end
    effector to ITS commanded pose — velocity and angular rate derived on the device (phystwin.py:131-138) — and the stepper's
    per-substep eef centres end there."""
    import torch
    from scipy.spatial.transform import Rotation
    from r2s_hip.rollout import BatchedRollout

    kw = dict(num_substeps=30, seed=7, n_env=2)
    a, b = BatchedRollout("tiny", **kw), BatchedRollout("tiny", **kw)
    for _ in range(3):
        a.step()
        b.step(b.synthetic_action(b.t))
    torch.cuda.synchronize()
    assert torch.equal(a.phys.x, b.phys.x) and torch.equal(a.out_color, b.out_color) and torch.equal(a.eef_xyz, b.eef_xyz)
    # (2) per-environment targets with a rotation
    xyz0, rot0 = a.eef_xyz.clone(), a.eef_rot.clone()
    d = torch.tensor([[0.002, -0.001, 0.0015], [-0.0015, 0.0005, -0.001]], device=a.device)
    rv = np.array([[0.0, 0.0, 0.02], [0.01, -0.015, 0.0]])
    rot_next = torch.from_numpy(np.stack([Rotation.from_rotvec(rv[e]).as_matrix().T @ rot0[e].cpu().numpy().astype(np.float64) for e in range(2)])).float().to(a.device)
    act = torch.cat([xyz0 + d, rot_next.reshape(2, 9), torch.tensor([[0.8], [0.6]], device=a.device)], 1)
    m = a.action13_to_motion(act)
    T = a.num_substeps * a.dt
    assert torch.allclose(m["eef_vel"] * T, d, atol=1e-7)
    assert np.allclose(m["eef_rot_vel"].cpu().numpy() * T, rv, atol=2e-6), (m["eef_rot_vel"].cpu().numpy() * T, rv)
    a.step(act)
    torch.cuda.synchronize()
    assert torch.allclose(a.eef_xyz, xyz0 + d, atol=1e-7) and torch.equal(a.eef_rot, rot_next)
    _, ctr, _, om = a.phys.mesh_motion(points=False)
    assert torch.allclose(ctr[:, -1], xyz0 + d, atol=2e-6)                    # the stepper's eef centre at the last substep
    assert torch.allclose(om, -0.5 * m["eef_rot_vel"], atol=1e-6)             # dynamic_omega = -eef_rot_vel / 2 (phystwin.py:451)
    assert bool(torch.isfinite(a.phys.x).all())


def test_observations_rerenders_a_sync_free_batch_that_overflowed():
    """the sync-free raster pipeline sizes its binning scratch from the PREVIOUS batch (+12.5 % + 4096); a batch
    that outgrows it loses its deepest instances.  The rollout must not hand such a frame to a closed-loop caller: the wrist cameras
    jump from 4 cm above the rope (most of the scene is nearer than z_threshold = 5 cm and culled: ~11 k instances) to 1.5 m above it
    (the whole table in view: ~20 k, beyond 11 k x 1.125 + 4096), `observations()` notices the overflow flag of that batch and renders
    the step again; what it returns equals the oracle's frame."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout("tiny", num_substeps=10, seed=9, n_env=2)
    rot = ro.eef_rot.clone()
    top = float(ro.ob["points"][:, 2].max())
    c = torch.from_numpy(ro.ob["points"].mean(0)).to(ro.device)

    def action(z):
        xyz_next = c[None].repeat(2, 1) + torch.from_numpy(ro.env_shift).to(ro.device)
        xyz_next[:, 2] = z
        return torch.cat([xyz_next, rot.reshape(2, 9), torch.ones(2, 1, device=ro.device)], 1)

    counts = []
    for _ in range(3):                       # lens inside the cull distance: few wrist instances; the capacity settles on this count
        ro.step(action(top + 0.04))
        ro.observations()
        counts.append(ro.last_num_rendered)
    assert ro.lossy_batches == 0
    ro.step(action(top + 1.5))              # 1.5 m up: the whole scene in view
    col, dep = ro.observations()
    torch.cuda.synchronize()
    counts.append(ro.last_num_rendered)
    assert ro.lossy_batches >= 1, f"the scenario must overflow the capacity of the previous batch (instance counts {counts})"
    for e in range(2):
        cam = ro.camera_numpy(e, 1)
        _, col_ref, _, dep_ref = oracle_render(ro.scene_numpy(e), cam)
        r = compare_images(col[e, 1].cpu().numpy(), dep[e, 1].cpu().numpy(), col_ref, dep_ref, what=f"re-rendered overflow batch, env {e}")
        assert r["frac_rgb"] <= 1e-3 and r["frac_depth"] <= 1e-3, (e, r, counts)
    # and the pipeline recovers: the next batches are sized for the new count
    ro.step(action(top + 1.5))
    ro.observations()
    assert ro.lossy_batches == 1, counts


@pytest.mark.gpu
def test_rot_to_quat_kernel_equals_the_torch_restatement_of_the_kornia_branch_scheme():
    """obs['robot']['eef_quat'] (env.py:62-66, phystwin.py:117): r2s_rot_to_quat against r2s_hip.rollout.rotation_matrix_to_quaternion (pinned against
    scipy up to the sign in tests/test_host_logic.py) on random rotations that hit all four branches, and on the identity."""
    import torch
    from scipy.spatial.transform import Rotation
    from r2s_hip.camera import rot_to_quat
    from r2s_hip.rollout import rotation_matrix_to_quaternion

    R = np.concatenate([Rotation.random(4000, random_state=3).as_matrix(), np.eye(3)[None],
                        Rotation.from_euler("xyz", [[179.9, 0, 0], [0, 179.9, 0], [0, 0, 179.9], [120, 120, 0]], degrees=True).as_matrix()]).astype(np.float32)
    Rt = torch.from_numpy(R).cuda()
    q_ref = rotation_matrix_to_quaternion(Rt)
    q = rot_to_quat(Rt)
    torch.cuda.synchronize()
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    assert (tr > 0).any() and (tr <= 0).sum() > 100, "every branch must be exercised"
    assert float((q - q_ref).abs().max()) < 2e-6, float((q - q_ref).abs().max())
    assert float((q.norm(dim=1) - 1).abs().max()) < 1e-5
