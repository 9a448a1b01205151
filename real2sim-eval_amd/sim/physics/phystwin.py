"""Drop-in for the reference's ``SpringMassDynamicsModule`` (sim/physics/phystwin.py:205-531): same constructor
arguments, same ``step(eef_xyz, eef_vel, eef_rot, eef_rot_vel, gripper_openness, eef_pts_func, init_eef_xyz)``,
``current_points`` / ``current_velocities``.  What the reference does on the host every step — read the collision forces
back, run the openness / grasp state machine, interpolate the finger vertices with scipy, build and upload the
[num_substeps, M, 3] vertex tensor (:367-513) — happens on the device here (r2s_phys_set_eef_table /
r2s_phys_set_eef_motion); the host only hands over the five small tensors it was given.

The PhysTwin case directory is read by ``r2s_hip.assets.load_phystwin_case`` (the reference needs open3d for the spring
search; this uses scipy's KD-tree with the same hybrid radius / k-nearest rule).  ``robot`` is the caller's
RobotPcSampler: only ``get_xarm_gripper_meshes(gripper_openness=1.0)`` / ``get_xarm_pusher_meshes()`` are called."""
from __future__ import annotations

import numpy as np
import torch

from r2s_hip import assets

from .spring_mass_warp import SpringMassSystemWarp


class SpringMassDynamicsModule:
    def __init__(self, phystwin_cfg, device, wp_device, case_name, data_path, zeroth_order_ckpt_path, first_order_ckpt_path, init_pts,
                 init_pose, static_meshes, robot, robot_type, use_pusher):
        phystwin_cfg.num_substeps = round(1.0 / phystwin_cfg.fps / phystwin_cfg.dt)               # :222
        self.device, self.wp_device, self.phystwin_cfg = device, wp_device, phystwin_cfg
        self.robot_type, self.use_pusher = robot_type, use_pusher
        pose = init_pose.detach().cpu().numpy() if isinstance(init_pose, torch.Tensor) else np.asarray(init_pose)
        case = assets.load_phystwin_case(data_path, zeroth_order_ckpt_path, first_order_ckpt_path, case_name, init_pose=pose,
                                         object_radius=phystwin_cfg.object_radius, object_max_neighbours=phystwin_cfg.object_max_neighbours)
        for key, value in case["params"].items():                                                   # :256-263
            assert hasattr(phystwin_cfg, key), key
            cur = getattr(phystwin_cfg, key)
            setattr(phystwin_cfg, key, int(value) if isinstance(cur, int) and not isinstance(cur, bool) else (float(value) if isinstance(cur, float) else value))
        t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt, device=device)  # noqa: E731
        self.init_pts = init_pts.to(torch.float32).to(device) if isinstance(init_pts, torch.Tensor) else t(init_pts)
        self.init_pts_aligned = t(case["points"])
        self.init_springs = t(case["springs"], torch.int32)
        self.init_rest_lengths = t(case["rest"])
        self.init_spring_Y = t(case["spring_Y"])
        self.collide_elas, self.collide_fric = t([case["collide_elas"]]), t([case["collide_fric"]])
        self.collide_self_elas, self.collide_self_fric = t([case["collide_self_elas"]]), t([case["collide_self_fric"]])
        if use_pusher:                                                                              # :305-306
            phystwin_cfg.collide_eef_fric = 0.2
        if robot is not None:                                                                       # :318-325
            dynamic_meshes = robot.get_xarm_pusher_meshes() if use_pusher else robot.get_xarm_gripper_meshes(gripper_openness=1.0)
        else:
            dynamic_meshes = []
        static_meshes = list(static_meshes or [])
        verts = [np.asarray(m.vertices if hasattr(m, "vertices") else m[0], np.float32).reshape(-1, 3) for m in dynamic_meshes]
        dynamic_vertices = t(np.concatenate(verts, axis=0)) if verts else torch.zeros(0, 3, device=device)
        self.simulator = SpringMassSystemWarp(                                                      # :336-357
            phystwin_cfg=phystwin_cfg, device=wp_device, init_vertices=self.init_pts_aligned, init_springs=self.init_springs,
            init_rest_lengths=self.init_rest_lengths, init_masses=torch.ones(len(self.init_pts_aligned), device=device),
            num_object_points=len(self.init_pts_aligned), init_spring_Y=torch.log(self.init_spring_Y).detach().clone(),
            collide_elas=self.collide_elas, collide_fric=self.collide_fric, collide_eef_elas=t([phystwin_cfg.collide_eef_elas]),
            collide_eef_fric=t([phystwin_cfg.collide_eef_fric]), collide_self_elas=self.collide_self_elas, collide_self_fric=self.collide_self_fric,
            init_collision_mask=None, init_velocities=None, dynamic_meshes=dynamic_meshes, static_meshes=static_meshes,
            dynamic_points=dynamic_vertices, use_pusher=use_pusher)
        self._table_of = None

    # the device keeps current_openness / grasped (phystwin.py:358-359); these read them back on demand
    @property
    def current_openness(self):
        return float(self.simulator._b.eef_state()[0][0]) if self._table_of is not None else None

    @property
    def grasped(self):
        return bool(self.simulator._b.eef_state()[1][0]) if self._table_of is not None else False

    def step(self, eef_xyz, eef_vel, eef_rot, eef_rot_vel, gripper_openness, eef_pts_func, init_eef_xyz):
        b = self.simulator._b
        if self.phystwin_cfg.self_collision:                                                        # :365-366
            self.simulator.update_collision_graph()
        if self._table_of is not eef_pts_func:  # the knots scipy's interp1d was built from (robot_pc_transformations.py:190)
            knots = np.asarray(eef_pts_func.y if hasattr(eef_pts_func, "y") else [eef_pts_func(k / 100.0) for k in range(101)], np.float64)
            if knots.shape[0] != len(getattr(eef_pts_func, "x", range(101))):
                knots = np.moveaxis(knots, -1, 0)  # interp1d stores the interpolation axis last
            b.set_eef_table(knots, np.asarray(init_eef_xyz.detach().cpu() if isinstance(init_eef_xyz, torch.Tensor) else init_eef_xyz, np.float32).reshape(-1)[:3],
                            float(self.phystwin_cfg.grasp_force_threshold))
            self._table_of = eef_pts_func
        op = None if self.use_pusher else gripper_openness.reshape(-1)[:1]
        b.set_eef_motion(eef_xyz[:1], eef_vel[:1], eef_rot[:1], eef_rot_vel[:1], op)            # first gripper, like :433, :436, :443
        if self.phystwin_cfg.use_graph:                                                             # :515-519
            self.simulator.graph.launch()
        else:
            self.simulator.step()
        return self.current_points

    @property
    def current_points(self):
        return self.simulator.wp_state.wp_x                                                         # :523-526

    @property
    def current_velocities(self):
        return self.simulator.wp_state.wp_v
