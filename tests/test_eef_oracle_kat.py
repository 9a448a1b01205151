"""Known-answer tests of the CPU restatement of the stepper's caller side (oracle/eef_oracle.py; reference
sim/physics/phystwin.py:362-513): openness / grasp state machine, finger interpolation, per-substep rigid motion."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from oracle.eef_oracle import EefOracle, axis_angle_to_rotation_matrix, make_eef_pts_func
from r2s_hip import synth

DT, N = 5e-5, 40


def _setup():
    tab, init, fl, fr = synth.gripper_eef_table()
    M = tab.shape[1]
    mesh_map = np.concatenate([np.zeros(len(fl), int), np.ones(len(fr), int)])
    return tab, init, make_eef_pts_func(tab), M, mesh_map


def _forces(mesh_map, left, right):
    f = np.zeros((len(mesh_map), 3), np.float32)
    l0, r0 = np.flatnonzero(mesh_map == 0)[0], np.flatnonzero(mesh_map == 1)[0]
    f[l0 + 18] = (left, 0, 0)
    f[r0 + 1] = (0, right, 0)
    return f


def test_kornia_axis_angle_restatement_matches_rodrigues_and_its_taylor_branch():
    aa = torch.tensor([[0.3, -0.2, 0.5], [0.0, 0.0, 1.2], [2e-4, -3e-4, 1e-4], [0.0, 0.0, 0.0]], dtype=torch.float32)
    R = axis_angle_to_rotation_matrix(aa).numpy()
    ref = Rotation.from_rotvec(aa.numpy().astype(np.float64)).as_matrix()
    assert np.abs(R[:2] - ref[:2]).max() < 2e-6           # Rodrigues branch (the 1e-6 in theta + eps costs ~1e-6)
    assert np.abs(R[2] - ref[2]).max() < 2e-7              # first-order branch: theta^2 = 1.4e-7 < 1e-6
    assert np.array_equal(R[3], np.eye(3, dtype=np.float32))


def test_pure_translation_keeps_the_finger_shape_and_halves_the_velocity():
    tab, init, fn, M, mesh_map = _setup()
    o = EefOracle(DT, N, 2000.0)
    xyz, vel = np.array([[0.4, 0.02, 0.3]], np.float32), np.array([[0.06, -0.03, 0.01]], np.float32)
    out = o.step(xyz, vel, np.eye(3, dtype=np.float32)[None], np.zeros((1, 3), np.float32), 0.7, fn, init, np.zeros((M * 0 + len(mesh_map), 3)), mesh_map)
    rel = synth.eef_world_points(fn(0.7), init, (0, 0, 0))
    for s in (0, N - 1):
        expect = xyz[0] + vel[0] * ((s + 1) * DT) + rel
        assert np.abs(out["interp_points"][s] - expect).max() < 2e-7
        assert np.abs(out["interp_center"][s] - (xyz[0] + vel[0] * ((s + 1) * DT))).max() < 1e-7
    assert out["dynamic_velocity"].shape == (2, 3) and np.allclose(out["dynamic_velocity"], vel * 0.5, atol=1e-8)
    assert np.allclose(out["dynamic_omega"], 0) and o.current_openness == float(np.float32(0.7)) and not o.grasped


def test_rotation_rate_turns_the_fingers_about_the_end_effector():
    tab, init, fn, M, mesh_map = _setup()
    o = EefOracle(DT, N, 2000.0)
    R0 = Rotation.from_euler("xyz", [0.2, -0.1, 0.4]).as_matrix().astype(np.float32)
    w = np.array([[0.0, 0.0, 3.0]], np.float32)
    out = o.step(np.array([[0.4, 0.0, 0.3]], np.float32), np.zeros((1, 3), np.float32), R0[None], w, 1.0, fn, init, np.zeros((len(mesh_map), 3)), mesh_map)
    s = N - 1
    # eef_rot_next = delta^T @ eef_rot (phystwin.py:379): the frame turns by MINUS the rate, hence dynamic_omega = -rate / 2
    Rn = Rotation.from_rotvec(w[0].astype(np.float64) * (s + 1) * DT).as_matrix().T @ R0.astype(np.float64)
    rel = synth.eef_world_points(fn(1.0), init, (0, 0, 0))
    expect = np.array([0.4, 0.0, 0.3]) + rel.astype(np.float64) @ Rn.T
    assert np.abs(out["interp_points"][s] - expect).max() < 1e-6
    assert np.allclose(out["dynamic_omega"], -w * 0.5)


def test_closing_moves_fingers_linearly_and_adds_the_closing_velocity():
    tab, init, fn, M, mesh_map = _setup()
    o = EefOracle(DT, N, 2000.0)
    z = np.zeros((1, 3), np.float32)
    I = np.eye(3, dtype=np.float32)[None]
    F0 = np.zeros((len(mesh_map), 3), np.float32)
    o.step(np.zeros((1, 3), np.float32), z, I, z, 1.0, fn, init, F0, mesh_map)
    out = o.step(np.zeros((1, 3), np.float32), z, I, z, 0.8, fn, init, F0, mesh_map)          # free closing: follows the command
    assert o.current_openness == float(np.float32(0.8)) and not o.grasped
    a, b = synth.eef_world_points(fn(1.0), init, (0, 0, 0)), synth.eef_world_points(fn(float(np.float32(0.8))), init, (0, 0, 0))
    assert np.abs(out["interp_points"][N - 1] - b).max() < 2e-7                                   # arrives at the new opening
    mid = N // 2 - 1
    assert np.abs(out["interp_points"][mid] - (a + (b - a) * ((mid + 1) / N))).max() < 2e-7
    half = M // 2
    v_close = (b - a) / (2 * DT * N)                                                                # "average velocity", :447
    assert np.allclose(out["dynamic_velocity"][0], v_close[:half].mean(0), atol=1e-4)
    assert np.allclose(out["dynamic_velocity"][1], v_close[half:].mean(0), atol=1e-4)
    assert out["dynamic_velocity"][0][1] > 0 > out["dynamic_velocity"][1][1]                      # the fingers approach each other


def test_grasp_state_machine_transitions():
    tab, init, fn, M, mesh_map = _setup()
    o = EefOracle(DT, N, 2000.0)
    z = np.zeros((1, 3), np.float32)
    I = np.eye(3, dtype=np.float32)[None]
    step = lambda cmd, fl, fr: o.step(z, z, I, z, cmd, fn, init, _forces(mesh_map, fl, fr), mesh_map)  # noqa: E731
    step(1.0, 0, 0)
    assert o.current_openness == 1.0 and not o.grasped
    step(0.6, 0, 0)                                            # closing, no contact: follow the command
    assert abs(o.current_openness - 0.6) < 1e-7 and not o.grasped
    step(0.4, 3000, 3000)                                      # both finger forces above the threshold: hold, grasp established
    assert abs(o.current_openness - 0.6) < 1e-7 and o.grasped
    step(0.2, 3000, 500)                                       # one finger below the threshold while grasped: creep by 0.05
    assert abs(o.current_openness - 0.55) < 1e-7 and o.grasped
    step(0.54, 500, 500)                                       # creep is limited by the command itself
    assert abs(o.current_openness - 0.54) < 1e-7 and o.grasped
    step(0.3, 50, 50)                                          # both forces small: grasp released, command followed
    assert abs(o.current_openness - 0.3) < 1e-7 and not o.grasped
    step(0.9, 3000, 3000)                                      # opening always follows the command; forces do not matter
    assert abs(o.current_openness - 0.9) < 1e-7 and not o.grasped
    out = step(-0.2, 0, 0)                                     # commands outside [0, 1] are clipped for the geometry only
    assert abs(o.current_openness + 0.2) < 1e-7
    assert np.abs(out["interp_points"][N - 1] - synth.eef_world_points(fn(0.0), init, (0, 0, 0))).max() < 2e-7


def test_pusher_branch_is_rigid_with_a_single_velocity_row():
    rod = synth.cylinder_mesh((0.0, 0.0, -0.1), radius=0.005, length=0.2, n_seg=16, n_rings=20)[0].astype(np.float64)
    init = np.array([0.3, 0.0, 0.4])
    rel = rod.copy(); rel[:, 1] *= -1; rel[:, 2] *= -1
    tab = np.repeat((init + rel)[None], 101, axis=0)
    fn = make_eef_pts_func(tab)
    o = EefOracle(DT, N, 2000.0, use_pusher=True)
    out = o.step(np.array([[0.5, 0.1, 0.3]], np.float32), np.array([[0.05, 0.0, 0.0]], np.float32), np.eye(3, dtype=np.float32)[None],
                 np.array([[0.0, 0.0, 0.5]], np.float32), None, fn, init.astype(np.float32))
    assert out["dynamic_velocity"].shape == (1, 3) and np.allclose(out["dynamic_velocity"], [[0.025, 0, 0]])
    assert o.current_openness == 1.0
    d0 = np.linalg.norm(out["interp_points"][0][:, None] - out["interp_points"][0][None], axis=-1)
    d1 = np.linalg.norm(out["interp_points"][N - 1][:, None] - out["interp_points"][N - 1][None], axis=-1)
    assert np.abs(d0 - d1).max() < 1e-6                         # rigid
