"""Drop-in operator surface of the reference's ``sim/physics/spring_mass_warp.py`` backed by the fused HIP
substep kernel of ``libr2s_hip.so`` (MI355X).  One object = one environment, like the reference; use
``r2s_hip.physics.PhysBatch`` directly to step many environments in one launch.

Kept for the caller ``sim/physics/phystwin.py`` (:336-357 constructor kwargs, :362-531 per-step use):
``SpringMassSystemWarp(...)``, ``.update_collision_graph()``, ``.set_mesh_interactive(...)``, ``.step()``,
``.graph`` (launch with ``.graph.launch()`` instead of ``wp.capture_launch(graph)``), ``.mesh_map.numpy()``,
``.collision_forces.numpy()``, ``.wp_state.wp_x`` / ``.wp_v`` (torch tensors — ``wp.to_torch`` becomes the
identity), ``.set_init_state``, ``.set_spring_Y``, ``.set_collide*``.  See INTEGRATION.md for the 4-line caller patch.
"""
from __future__ import annotations

import numpy as np
import torch

from r2s_hip.physics import PhysBatch


class _HostArray:
    """Quacks like a ``wp.array`` for the two read-backs the caller does (``.numpy()``, phystwin.py:383-386)."""

    def __init__(self, getter):
        self._get = getter

    def numpy(self):
        a = self._get()
        return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)

    def __array__(self, dtype=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a


class State:
    """``State`` of the reference (:8-17): ``wp_x`` / ``wp_v`` are [N,3] float32 device tensors that are
    refreshed in place after every step (the reference exposes zero-copy views of its own buffers)."""

    def __init__(self, batch: PhysBatch):
        self._b = batch

    @property
    def wp_x(self):
        return self._b.x[0]

    @property
    def wp_v(self):
        return self._b.v[0]


class _Graph:
    """Stands in for the captured CUDA graph (:723-726): ``graph.launch()`` == ``wp.capture_launch(graph)``."""

    def __init__(self, sim):
        self._sim = sim

    def launch(self):
        self._sim._launch_graph()


def _scalar(v):
    return float(np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, dtype=np.float32).reshape(-1)[0])


class SpringMassSystemWarp:
    def __init__(self, phystwin_cfg, device, init_vertices, init_springs, init_rest_lengths, init_masses,
                 num_object_points, init_spring_Y=None, collide_elas=None, collide_fric=None, collide_eef_elas=None,
                 collide_eef_fric=None, collide_self_elas=None, collide_self_fric=None, init_collision_mask=None,
                 init_velocities=None, dynamic_meshes=None, static_meshes=None, dynamic_points=None, use_pusher=False):
        cfg = phystwin_cfg
        self.device = device
        self.dt = cfg.dt
        self.num_substeps = cfg.num_substeps
        self.dashpot_damping = cfg.dashpot_damping
        self.drag_damping = cfg.drag_damping
        self.reverse_factor = 1.0 if not cfg.reverse_z else -1.0
        self.spring_Y_min = cfg.spring_Y_min
        self.spring_Y_max = cfg.spring_Y_max
        self.self_collision = cfg.self_collision
        self.use_pusher = use_pusher
        self.collision_dist = cfg.collision_dist
        self.n_springs = int(init_springs.shape[0])
        self.num_object_points = int(num_object_points)
        assert num_object_points == init_vertices.shape[0]  # :526
        has_mesh = bool(dynamic_meshes) or bool(static_meshes)
        if has_mesh:
            assert isinstance(dynamic_points, torch.Tensor)  # :696
            self.num_eefs = (len(dynamic_meshes or []) // 2) if not use_pusher else len(dynamic_meshes or [])
            assert self.num_eefs <= 1  # :698
        if init_spring_Y is None:  # :586-590 default: log(cfg.init_spring_Y) for every spring
            init_spring_Y = torch.full((self.n_springs,), float(np.log(np.float32(cfg.init_spring_Y))), dtype=torch.float32)
        pick = lambda given, name: getattr(cfg, name) if given is None else _scalar(given)  # noqa: E731
        dev = torch.device(str(device).replace("cuda", "cuda") if isinstance(device, str) else device)
        self._b = PhysBatch(
            init_vertices=init_vertices[None] if init_vertices.ndim == 2 else init_vertices,
            init_springs=init_springs, init_rest_lengths=init_rest_lengths, init_masses=init_masses[:num_object_points],
            init_spring_Y=init_spring_Y, dt=cfg.dt, num_substeps=cfg.num_substeps, dashpot_damping=cfg.dashpot_damping,
            drag_damping=cfg.drag_damping, spring_Y_min=cfg.spring_Y_min, spring_Y_max=cfg.spring_Y_max,
            collision_dist=cfg.collision_dist, reverse_z=cfg.reverse_z, self_collision=cfg.self_collision,
            collide_elas=pick(collide_elas, "collide_elas"), collide_fric=pick(collide_fric, "collide_fric"),
            collide_eef_elas=pick(collide_eef_elas, "collide_eef_elas"), collide_eef_fric=pick(collide_eef_fric, "collide_eef_fric"),
            collide_self_elas=pick(collide_self_elas, "collide_self_elas"), collide_self_fric=pick(collide_self_fric, "collide_self_fric"),
            init_collision_mask=init_collision_mask, init_velocities=None if init_velocities is None else init_velocities[None, :num_object_points],
            dynamic_meshes=dynamic_meshes, static_meshes=static_meshes, use_pusher=use_pusher, device=dev)
        self.wp_state = State(self._b)
        self.all_meshes_warp = object() if has_mesh else None
        if has_mesh:
            self.mesh_map = _HostArray(lambda: self._b.mesh_map)
            self.face_map = _HostArray(lambda: self._b.face_map)
            self.collision_forces = _HostArray(lambda: self._b.collision_forces()[0])
            self.num_dynamic_points = len(dynamic_points)
            self.num_dynamic_velocities = self.num_eefs * 2 if not use_pusher else self.num_eefs
        self.graph = _Graph(self) if getattr(cfg, "use_graph", True) else None

    # -- reference methods --------------------------------------------------------------------------------
    def create_resting_case(self):  # :729-740
        self._b.create_resting_case()

    def set_init_state(self, x, v):  # :742-767
        self._b.set_state(x[None], None if v is None else v[None])

    def set_mesh_interactive(self, interpolated_dynamic_points, interpolated_center, dynamic_velocity, dynamic_omega):  # :769-804
        self._b.set_mesh_interactive(interpolated_dynamic_points[None], interpolated_center.reshape(1, self.num_substeps, 3),
                                     dynamic_velocity[None], dynamic_omega[None])

    def update_collision_graph(self):  # :806-821
        assert self.self_collision
        self._b.update_collision_graph()

    def step(self):  # :823-943
        self._b.step(0, 0)

    def _launch_graph(self):
        self._b.step(0, 0)

    def set_spring_Y(self, spring_Y):  # :946-953 (log stiffness)
        self._b.set_spring_Y(spring_Y)

    def set_collide(self, collide_elas, collide_fric):  # :955-967
        self._b.set_params(collide_elas=collide_elas, collide_fric=collide_fric)

    def set_collide_eef(self, collide_eef_elas, collide_eef_fric):  # :969-981
        self._b.set_params(collide_eef_elas=collide_eef_elas, collide_eef_fric=collide_eef_fric)

    def set_collide_self(self, collide_self_elas, collide_self_fric):  # :983-995
        self._b.set_params(collide_self_elas=collide_self_elas, collide_self_fric=collide_self_fric)
