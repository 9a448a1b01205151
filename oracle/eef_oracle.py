"""CPU restatement of the caller side of the PhysTwin stepper: what ``SpringMassDynamicsModule.step`` computes before it
calls ``set_mesh_interactive`` (reference sim/physics/phystwin.py:362-513) — gripper openness / grasp state machine,
finger-vertex interpolation, per-substep rigid motion of the dynamic collision meshes.

TEST INFRASTRUCTURE ONLY: imported by tests/ and oracle/parity_gate.py (and nothing else) as the checker of the HIP kernels
k_eef_prepare / k_eef_points.  PINNED (round 4): tests/golden/eef_step.npz holds inputs and outputs of the REFERENCE's own
``SpringMassDynamicsModule.step`` — phystwin.py imported with tests/golden/warp_shim.py standing in for warp, placeholder modules
for open3d / sapien / transforms3d / urdfpy, the module object created without ``__init__`` and a recorder in place of the warp
simulator (tests/golden/make_eef_golden.py) — over scripted open -> closing -> grasp -> held -> creep -> release sequences of the
gripper branch (40 and 667 substeps) and the pusher branch; this restatement reproduces the fixture BIT FOR BIT
(tests/test_eef_reference_fixture.py), the device kernels within 1e-6 m / state machine exact (tests/test_eef_gpu.py).
Still unpinned: kornia's ``axis_angle_to_rotation_matrix`` (third party, not installed; restated from its published source here,
in the fixture's generator and in the kernel).  The torch / numpy / scipy operations of the cited lines are re-issued in the same
order on the CPU (same dtypes: python float64 for the state machine, scipy interp1d for the vertices, float32 torch ops for the
motion).
"""
from __future__ import annotations

import numpy as np
import scipy.interpolate
import torch


def axis_angle_to_rotation_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.conversions.axis_angle_to_rotation_matrix restated: (N,3) -> (N,3,3)."""
    aa = axis_angle
    theta2 = (aa[:, None, :] @ aa[:, :, None]).reshape(-1, 1)
    theta = torch.sqrt(theta2)
    w = aa / (theta + 1e-6)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    k = 1.0 - c
    normal = torch.cat([c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k,
                        wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
                        -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k], dim=1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    mask = (theta2 > 1e-6).view(-1, 1, 1)
    return torch.where(mask, normal, taylor)


def make_eef_pts_func(eef_pts_list):
    """robot_pc_transformations.py:190 / :225."""
    n = len(eef_pts_list)
    return scipy.interpolate.interp1d(np.arange(n) / (n - 1.0), np.asarray(eef_pts_list), axis=0)


class EefOracle:
    """One environment.  ``step`` returns what the reference hands to set_mesh_interactive."""

    def __init__(self, dt, num_substeps, grasp_force_threshold, use_pusher=False):
        self.dt, self.n, self.thr, self.use_pusher = float(dt), int(num_substeps), grasp_force_threshold, bool(use_pusher)
        self.current_openness = None
        self.grasped = False

    def step(self, eef_xyz, eef_vel, eef_rot, eef_rot_vel, gripper_openness, eef_pts_func, init_eef_xyz, collision_forces=None, mesh_map=None):
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        eef_xyz, eef_vel, eef_rot, eef_rot_vel = t(eef_xyz).reshape(-1, 3), t(eef_vel).reshape(-1, 3), t(eef_rot).reshape(-1, 3, 3), t(eef_rot_vel).reshape(-1, 3)
        init_eef_xyz = t(init_eef_xyz).reshape(-1, 3)
        n_grippers = eef_xyz.shape[0]
        n_substeps = self.n
        dts = torch.linspace(1, n_substeps, n_substeps) * self.dt                                     # :374
        eef_xyz_next = eef_xyz[None] + eef_vel[None] * dts[:, None, None]                             # :376
        eef_aa_delta = eef_rot_vel[None] * dts[:, None, None]                                         # :377
        eef_rot_delta = axis_angle_to_rotation_matrix(eef_aa_delta.reshape(-1, 3)).reshape(n_substeps, n_grippers, 3, 3)
        eef_rot_next = eef_rot_delta.permute(0, 1, 3, 2) @ eef_rot                                    # :379
        if not self.use_pusher:
            openness = float(np.float32(gripper_openness))                                            # .item(), :369
            if self.current_openness is None:
                self.current_openness = openness
            force = np.asarray(collision_forces, np.float32)
            mesh_map = np.asarray(mesh_map)
            left, right = force[mesh_map == 0], force[mesh_map == 1]                                  # :382-387
            lf = left[18] + left[19] + left[1]
            rf = right[18] + right[19] + right[1]
            norm = np.linalg.norm(np.stack([lf, rf], axis=0), axis=1)                                  # :391-392
            openness_before = self.current_openness
            if np.all(norm < 100):
                self.grasped = False
            if openness < self.current_openness:
                if np.all(norm > self.thr):
                    openness = self.current_openness
                    self.grasped = True
                elif self.grasped:
                    self.current_openness = max(openness, self.current_openness - 0.05)
                    openness = self.current_openness
                else:
                    self.current_openness = openness
            else:
                self.current_openness = openness
            assert self.current_openness == openness
            openness = np.clip(openness, 0.0, 1.0)
            openness_before = np.clip(openness_before, 0.0, 1.0)
        else:
            self.current_openness = 1.0
            openness = openness_before = 1.0
        eef_pts = torch.from_numpy(np.asarray(eef_pts_func(openness))).to(torch.float32)               # :416-417
        eef_pts_before = torch.from_numpy(np.asarray(eef_pts_func(openness_before))).to(torch.float32)
        eef_pts_delta = eef_pts - eef_pts_before
        eef_pts_delta[:, 1] *= -1
        eef_pts_delta[:, 2] *= -1
        relative_eef_pts = eef_pts_before - init_eef_xyz
        relative_eef_pts[:, 1] *= -1
        relative_eef_pts[:, 2] *= -1
        relative_eef_pts = relative_eef_pts[None, None, :, :]
        relative_eef_pts = relative_eef_pts + eef_pts_delta[None, None] / (self.dt * n_substeps) * dts[:, None, None, None]
        pts = eef_xyz_next[:, :, None] + relative_eef_pts @ eef_rot_next.permute(0, 1, 3, 2)           # :432
        pts = pts[:, 0]
        center = eef_xyz_next[:, 0]
        dynamic_velocity = eef_vel[0] * 0.5
        if not self.use_pusher:
            d = eef_pts_delta @ eef_rot[0].permute(1, 0)
            closing = d / (2 * self.dt * n_substeps)
            half = len(closing) // 2
            closing = torch.stack([closing[:half].mean(0), closing[half:].mean(0)], dim=0)
            dynamic_velocity = dynamic_velocity + closing                                              # (2, 3)
        else:
            dynamic_velocity = dynamic_velocity[None]                                                  # (1, 3)
        dynamic_omega = (-eef_rot_vel[0] * 0.5)[None]
        return dict(interp_points=pts.numpy(), interp_center=center.numpy(), dynamic_velocity=dynamic_velocity.numpy(),
                    dynamic_omega=dynamic_omega.numpy())
