"""The C restatement of the stepper (oracle/physics_oracle*.c) against fixtures produced by EXECUTING THE REFERENCE's own
kernel bodies (tests/golden/physics_kernels.npz <- tests/golden/make_physics_golden.py + warp_shim.py): spring forces with
the stiffness gate, velocity update, self-collision impulses on given candidate lists, mesh-collision response on given
query answers (gripper fingers + static box, re-query, per-face forces), ground contact with time of impact — chained in
the order of SpringMassSystemWarp.step.  Differences are float32 summation order only (the reference accumulates spring
forces in spring order with atomics, the restatement in the same spring order; the shim never contracts to FMA)."""
import os

import numpy as np

import oracle

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "physics_kernels.npz"))
TOL = 2e-6  # metres / (m/s): float32 rounding over a dozen substeps


def test_springs_gate_velocity_and_ground_contact_trajectory():
    o = oracle.PhysOracle(G["A_x0"], G["A_springs"], G["A_rest"], G["A_logY"], v0=G["A_v0"], num_substeps=1, spring_Y_min=float(G["A_Ymin"]),
                          self_collision=False)
    gated = np.exp(G["A_logY"]) <= G["A_Ymin"]
    assert gated.any() and not gated.all()
    for k in range(len(G["A_x_traj"])):
        o.step(1, 0)
        assert np.abs(o.x - G["A_x_traj"][k]).max() < TOL, k
        assert np.abs(o.v - G["A_v_traj"][k]).max() < 2e-4, k          # velocities ~1 m/s: 1e-4 relative
    assert (G["A_v_traj"][-1][:, 2] > 0).any(), "the fixture contains a ground bounce"
    # first substep in isolation: velocity after the force kernel (no contact yet for most particles)
    o2 = oracle.PhysOracle(G["A_x0"], G["A_springs"], G["A_rest"], G["A_logY"], v0=G["A_v0"], num_substeps=1, spring_Y_min=float(G["A_Ymin"]),
                           self_collision=False, drag_damping=3.0)
    o2.step(1, 0)
    free = G["A_x0"][:, 2] + G["A_v_after_force"][:, 2] * 5e-5 > 1e-6      # particles that do not reach the floor this substep
    assert free.sum() > 10 and np.abs(o2.v[free] - G["A_v_after_force"][free]).max() < 2e-4


def test_self_collision_impulses_on_given_candidate_lists():
    o = oracle.PhysOracle(G["B_x0"], G["B_springs"], G["B_rest"], G["B_logY"], v0=G["B_v0"], num_substeps=1, self_collision=True)
    o.coll_idx[:, : G["B_coll_idx"].shape[1]] = G["B_coll_idx"]
    o.coll_num[:] = G["B_coll_num"]
    assert G["B_coll_num"].sum() > 100
    for k in range(len(G["B_x_traj"])):
        o.step(1, 0)
        assert np.abs(o.x - G["B_x_traj"][k]).max() < TOL, k
        assert np.abs(o.v - G["B_v_traj"][k]).max() < 5e-4, k
    moved = np.abs(G["B_v_traj"][-1][:, 0] - G["B_v0"][:, 0]).max()
    assert moved > 0.5, "the blobs exchanged momentum in the fixture"


def test_mesh_collision_response_gripper_and_static_with_forces():
    n_dyn = int(G["C_n_dyn"])
    verts, faces, mm = G["C_verts"], G["C_faces"], G["C_mesh_map"]
    nl = int((mm == 0).sum()); nr = int((mm == 1).sum())
    vl = int(faces[:nl].max()) + 1
    dyn = [(verts[:vl], faces[:nl]), (verts[vl:n_dyn], faces[nl:nl + nr] - vl)]
    sta = [(verts[n_dyn:], faces[nl + nr:] - n_dyn)]
    n_sub = len(G["C_x_traj"])
    o = oracle.PhysOracle(G["C_x0"], G["C_springs"], G["C_rest"], G["C_logY"], v0=G["C_v0"], num_substeps=n_sub, self_collision=False,
                          dynamic_meshes=dyn, static_meshes=sta, collide_eef_elas=0.5, collide_eef_fric=1.0)
    assert np.array_equal(o.mesh_map, mm)
    o.set_mesh_interactive(G["C_interp"], G["C_centers"], G["C_dyn_vel"], G["C_dyn_omega"])
    seen_force = 0.0
    for k in range(n_sub):
        o.step(1, k)
        assert np.abs(o.x - G["C_x_traj"][k]).max() < TOL, k
        assert np.abs(o.v - G["C_v_traj"][k]).max() < 2e-3, k           # contact velocities reach ~6 m/s
        ref_f = G["C_forces_traj"][k]
        seen_force = max(seen_force, float(np.abs(ref_f).max()))
        assert np.allclose(o.collision_forces, ref_f, rtol=2e-3, atol=1e-3 * max(1.0, np.abs(ref_f).max())), k
    assert seen_force > 1e4
