"""The drop-in ``sim.physics.phystwin.SpringMassDynamicsModule`` end to end: a PhysTwin case directory on disk (files and
keys of phystwin.py:231-298), a stand-in robot that only provides the two finger meshes, the reference's ``step``
signature — against the CPU oracles (oracle.eef_oracle for the caller side, oracle.PhysOracle for the stepper)."""
import os
import pickle as pkl
from types import SimpleNamespace

import numpy as np
import pytest

from util_physics import DEFAULTS
from util_parity import close

pytestmark = pytest.mark.gpu


def test_dynamics_module_from_case_directory_matches_the_oracles(tmp_path):
    import torch
    import oracle
    from oracle.eef_oracle import EefOracle, make_eef_pts_func
    from r2s_hip import synth
    from sim.physics.phystwin import SpringMassDynamicsModule

    ob = synth.phystwin_object("sloth", 500, 6)
    pts = ob["points"].astype(np.float64)
    case = "demo"
    for d in ("data", "zeroth", "first"):
        os.makedirs(tmp_path / d / case / ("train" if d == "first" else ""), exist_ok=True)
    with open(tmp_path / "data" / case / "final_data.pkl", "wb") as f:
        pkl.dump(dict(object_points=pts[None, :300], object_colors=np.zeros((1, 300, 3)), surface_points=pts[300:400], interior_points=pts[400:]), f)
    with open(tmp_path / "zeroth" / case / "optimal_params.pkl", "wb") as f:
        pkl.dump(dict(global_spring_Y=3000.0, collide_object_elas=0.5, collide_object_fric=0.3), f)
    springs, rest = synth.build_springs(pts)
    Y = np.exp(ob["log_Y"]).astype(np.float32)
    torch.save(dict(spring_Y=torch.cat([torch.from_numpy(Y), torch.ones(5)]), collide_elas=torch.tensor([0.5]), collide_fric=torch.tensor([0.3]),
                    collide_object_elas=torch.tensor([0.5]), collide_object_fric=torch.tensor([0.3]), num_object_springs=len(springs)),
               tmp_path / "first" / case / "train" / "best_3.pth")
    n_sub = 40
    cfg = SimpleNamespace(**dict(DEFAULTS, fps=1.0 / (5e-5 * n_sub), init_spring_Y=3e4, use_graph=True, collision_requires_grad=False, object_radius=0.02,
                                 object_max_neighbours=30, grasp_force_threshold=2000.0, self_collision=False, collide_eef_elas=0.5))
    tab, init, fl, fr = synth.gripper_eef_table()
    fn = make_eef_pts_func(tab)
    c = pts.mean(0); top = pts[:, 2].max()
    eef0 = np.array([c[0], c[1], top + 0.085], np.float32)
    w0 = synth.eef_world_points(fn(1.0), init, eef0)
    M = len(w0) // 2
    robot = SimpleNamespace(get_xarm_gripper_meshes=lambda gripper_openness=1.0: [SimpleNamespace(vertices=w0[:M], triangles=fl), SimpleNamespace(vertices=w0[M:], triangles=fr)])
    mod = SpringMassDynamicsModule(cfg, "cuda:0", "cuda:0", case, str(tmp_path / "data"), str(tmp_path / "zeroth"), str(tmp_path / "first"),
                                   init_pts=torch.from_numpy(pts.astype(np.float32)), init_pose=torch.eye(4), static_meshes=[], robot=robot,
                                   robot_type="xarm7", use_pusher=False)
    assert cfg.num_substeps == n_sub and cfg.init_spring_Y == 3000.0 and mod.current_openness is None
    assert np.array_equal(mod.init_springs.cpu().numpy(), springs)
    o = oracle.PhysOracle(pts.astype(np.float32), springs, rest, ob["log_Y"], num_substeps=n_sub, self_collision=False, collide_eef_elas=0.5,
                          dynamic_meshes=[(w0[:M], fl), (w0[M:], fr)])
    eo = EefOracle(5e-5, n_sub, 2000.0)
    xyz = eef0[None].copy()
    rot = np.eye(3, dtype=np.float32)[None]
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    touched = False
    for k, (vz, cmd) in enumerate([(-10.0, 1.0), (-10.0, 0.7), (-4.0, 0.5)]):
        vel = np.array([[0.0, 0.0, vz]], np.float32); rv = np.array([[0.0, 0.0, 0.2]], np.float32)
        ref = eo.step(xyz, vel, rot, rv, cmd, fn, init, o.collision_forces, o.mesh_map)
        o.set_mesh_interactive(ref["interp_points"], ref["interp_center"], ref["dynamic_velocity"], ref["dynamic_omega"])
        o.step()
        x = mod.step(tt(xyz), tt(vel), tt(rot), tt(rv), torch.tensor([[cmd]], dtype=torch.float32).cuda(), fn, tt(init))
        assert x.shape == (len(pts), 3) and np.abs(x.cpu().numpy() - o.x).max() < 1e-5, k
        assert close(mod.current_velocities.cpu().numpy(), o.v, 5e-3)
        assert mod.current_openness == eo.current_openness and mod.grasped == eo.grasped
        touched = touched or np.abs(o.collision_forces).max() > 0 or np.abs(o.x - pts).max() > 2e-4
        xyz = xyz + vel * (n_sub * 5e-5)
    assert touched, "the fingers reached the object in this scenario"
