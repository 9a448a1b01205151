"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.  The reference's control loop around its stepper, closed on the CPU:

    forces of the last substep  ->  grasp state machine + finger kinematics  ->  set_mesh_interactive  ->  667 substeps  ->  forces

(phystwin.py:362-521: ``SpringMassDynamicsModule.step`` reads ``simulator.collision_forces`` of the PREVIOUS step, decides the opening,
interpolates the finger vertices, hands them to the simulator and launches the graph.)  ``oracle.PhysOracle`` is the stepper,
``oracle.eef_oracle.EefOracle`` its caller; this file only wires the two together on one of the synthetic scenes
(``r2s_hip.rollout.scene_setup`` — plain numpy, no device) with the scene's action trace, so that a rollout THROUGH a grasp —
approach, closing, grasp detected, opening frozen / creeping, lift, release — exists as a checker next to the HIP rollout
(tests/test_grasp_closed_loop_gpu.py) and on its own (tests/test_closed_loop_oracle.py: the state machine really does reach
``grasped`` from the stepper's own forces).  Imported by tests/ and tools/ only.
"""
from __future__ import annotations

import numpy as np


class OracleRollout:
    """One environment of a synthetic scene, stepped like ``BatchedRollout.step`` steps its batch: candidate rebuild, caller
    (state machine + kinematics, from the previous step's forces), ``num_substeps`` substeps.  ``x`` / ``v`` may be overwritten by
    the caller between steps (a test copies the device state in after settling)."""

    def __init__(self, scn, env_shift=None, threads=1, self_collision=True, grasp_force_threshold=3e4, f64=False):
        from . import PhysOracle, set_threads
        from .eef_oracle import EefOracle, make_eef_pts_func

        self.scn = scn
        ob = scn["ob"]
        sh = np.zeros(3, np.float32) if env_shift is None else np.asarray(env_shift, np.float32)
        dyn = [(v + sh, f) for v, f in scn["dyn"]]
        self.phys = PhysOracle(ob["points"] + sh, ob["springs"], ob["rest"], ob["log_Y"], num_substeps=scn["num_substeps"],
                               self_collision=self_collision, dynamic_meshes=dyn, static_meshes=scn["sta"] or None,
                               use_pusher=scn["use_pusher"], collide_eef_fric=0.2 if scn["use_pusher"] else 1.0, f64=f64)
        self.eef = EefOracle(scn["dt"], scn["num_substeps"], grasp_force_threshold, use_pusher=scn["use_pusher"])
        self.fn = make_eef_pts_func(scn["eef_table"])
        self.eef_xyz = (scn["eef0"] + sh).astype(np.float32)[None]
        self.eef_rot = np.eye(3, dtype=np.float32)[None]
        self.eef_rot_vel = np.zeros((1, 3), np.float32)
        self.t = 0
        self.threads = int(threads)
        set_threads(max(1, self.threads))
        self.log = []

    @property
    def x(self):
        return self.phys.x

    @property
    def v(self):
        return self.phys.v

    def filtered_forces(self):
        """|sum of the forces on faces 18, 19, 1| of each finger after the last substep (phystwin.py:386-392)."""
        F, mm = np.asarray(self.phys.collision_forces, np.float32), self.phys.mesh_map
        out = []
        for m in (0, 1):
            f = F[mm == m]
            out.append(float(np.linalg.norm(f[18] + f[19] + f[1])) if len(f) > 19 else 0.0)
        return out

    def begin_step(self, vel=None, openness=None, forces=None):
        """The caller side of one env step (phystwin.py:362-460): candidate rebuild, grasp state machine on the forces the PREVIOUS step's
        last substep left (``forces``: another stepper's instead of this one's own — a test that wants both state machines on equal
        input), finger kinematics, set_mesh_interactive.  Returns what was handed to the stepper."""
        from r2s_hip.rollout import eef_velocity, open_command

        scn, ph = self.scn, self.phys
        if ph.self_collision:
            ph.update_collision_graph()
        self._vel = eef_velocity(scn, self.t) if vel is None else np.asarray(vel, np.float32)
        op = None if scn["use_pusher"] else (open_command(scn, self.t) if openness is None else float(openness))
        F = np.asarray(ph.collision_forces if forces is None else forces, np.float32)
        self._norms = self.filtered_forces() if not scn["use_pusher"] else [0.0, 0.0]
        self._op = op
        ref = self.eef.step(self.eef_xyz, self._vel[None], self.eef_rot, self.eef_rot_vel, op, self.fn, scn["eef_init"], F, ph.mesh_map)
        ph.set_mesh_interactive(ref["interp_points"], ref["interp_center"], ref["dynamic_velocity"], ref["dynamic_omega"])
        return ref

    def run(self, n_substeps, first_substep=0):
        """Substeps [first, first + n) of the env step begun with ``begin_step``."""
        from . import phys_step_batch_par

        ph = self.phys
        if self.threads > 1 and not ph.f64:   # (positions / velocities equal the sequential stepper's bit for bit under the checker build)
            phys_step_batch_par([ph], int(n_substeps), self.threads, first_substep=int(first_substep))
        else:
            ph.step(int(n_substeps), int(first_substep))

    def end_step(self):
        scn, ph = self.scn, self.phys
        self.eef_xyz = (self.eef_xyz + self._vel[None] * np.float32(scn["num_substeps"] * scn["dt"])).astype(np.float32)
        self.log.append(dict(t=self.t, command=self._op, openness=self.eef.current_openness, grasped=bool(self.eef.grasped), force_in=self._norms,
                             candidates=int((ph.coll_num > 0).sum()) if ph.self_collision else 0,
                             hits=int((np.abs(ph.collision_forces).sum(1) > 0).sum())))
        self.t += 1

    def step(self, vel=None, openness=None):
        """One env step of the scene's action trace (or of the given end-effector velocity [3] / commanded opening)."""
        ref = self.begin_step(vel, openness)
        self.run(self.phys.num_substeps, 0)
        self.end_step()
        return ref
