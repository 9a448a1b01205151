"""Python host of the physics C ABI (include/r2s_physics.h): a batch of environments that share one
PhysTwin, stepped by the fused HIP substep kernel.  torch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, cur_stream


class R2SPhysParams(C.Structure):
    _fields_ = [
        ("dt", C.c_float), ("dashpot_damping", C.c_float), ("drag_damping", C.c_float), ("spring_Y_min", C.c_float),
        ("spring_Y_max", C.c_float), ("collision_dist", C.c_float), ("collide_elas", C.c_float), ("collide_fric", C.c_float),
        ("collide_eef_elas", C.c_float), ("collide_eef_fric", C.c_float), ("collide_self_elas", C.c_float),
        ("collide_self_fric", C.c_float), ("reverse_z", C.c_int32), ("self_collision", C.c_int32), ("use_pusher", C.c_int32),
        ("num_substeps", C.c_int32),
    ]


class R2SPhysDesc(C.Structure):
    _fields_ = [
        ("params", R2SPhysParams), ("n_env", C.c_int32), ("num_object_points", C.c_int32), ("num_springs", C.c_int32),
        ("init_vertices", C.c_void_p), ("init_velocities", C.c_void_p), ("init_springs", C.c_void_p),
        ("init_rest_lengths", C.c_void_p), ("init_spring_Y", C.c_void_p), ("init_masses", C.c_void_p),
        ("init_collision_mask", C.c_void_p), ("n_dynamic_meshes", C.c_int32), ("n_static_meshes", C.c_int32),
        ("mesh_num_vertices", C.c_void_p), ("mesh_num_faces", C.c_void_p), ("mesh_vertices", C.c_void_p),
        ("mesh_triangles", C.c_void_p), ("collision_capacity", C.c_int32),
    ]


class R2SFlavourIn(C.Structure):
    """include/r2s_physics.h: the input of the flavour selection (counters of env step t - lag — 2, small batches 1 —, capabilities, switches)."""
    _fields_ = [(n, C.c_int32) for n in (
        "have_counters", "near_mesh", "query_needed", "servers_ran_out", "srv_exhausted", "n_candidates", "n_substeps", "full_step",
        "n_faces", "any_large", "block", "split_ok", "resident_ok", "srv_ok", "pf_ok", "has_vx", "self_collision", "n_blocks", "n_env",
        "n_cu", "srv_wg_cap", "resident_pref", "res_self", "res_self_srv", "pf_pref", "force_defer", "chains_override", "srv_own", "srv_quad")]


class R2SFlavourOut(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "variant", "mesh", "mesh_defer", "resident", "self_srv", "pf", "contact_finish", "chains", "n_srv", "srv_quad", "srv_own",
        "srv_exhausted", "graph_slot", "sum_class")] + [("kernel", C.c_char * 192)]


def pick_flavour(**fields):
    """The flavour selection as the pure function it is (r2s_phys_debug_pick_flavour; no GPU, no handle): keyword arguments are the
    fields of ``R2SFlavourIn`` (unset: 0, ``force_defer`` / ``srv_quad``: -1 = automatic); returns the fields of ``R2SFlavourOut``."""
    fin = R2SFlavourIn()
    fin.force_defer, fin.srv_quad = -1, -1
    for k, v in fields.items():
        if not hasattr(fin, k):
            raise KeyError(k)
        setattr(fin, k, int(v))
    out = R2SFlavourOut()
    check(_bind().r2s_phys_debug_pick_flavour(C.byref(fin), C.byref(out)), "r2s_phys_debug_pick_flavour")
    d = {n: int(getattr(out, n)) for n, _ in R2SFlavourOut._fields_ if n != "kernel"}
    d["kernel"] = out.kernel.decode()
    return d


_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if _bound:
        return L
    vp, i32 = C.c_void_p, C.c_int
    L.r2s_phys_create.restype = i32
    L.r2s_phys_create.argtypes = [C.POINTER(R2SPhysDesc), C.POINTER(vp), vp]
    L.r2s_phys_destroy.restype = None
    L.r2s_phys_destroy.argtypes = [vp]
    for name, args in dict(
        r2s_phys_set_state=[vp, vp, vp, vp], r2s_phys_get_state=[vp, vp, vp, vp], r2s_phys_create_resting_case=[vp, vp],
        r2s_phys_update_collision_graph=[vp, vp], r2s_phys_set_mesh_interactive=[vp, vp, vp, vp, vp, vp],
        r2s_phys_step=[vp, i32, i32, vp], r2s_phys_collision_forces=[vp, C.POINTER(vp), C.POINTER(C.c_int32)],
        r2s_phys_mesh_maps=[vp, vp, vp], r2s_phys_collision_lists=[vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32)],
        r2s_phys_collision_max_count=[vp, C.POINTER(C.c_int32), vp], r2s_phys_set_spring_Y=[vp, vp, vp],
        r2s_phys_set_eef_table=[vp, C.c_int32, vp, vp, C.c_float, vp], r2s_phys_set_eef_motion=[vp, vp, vp, vp, vp, vp, vp],
        r2s_phys_eef_state=[vp, C.POINTER(vp), C.POINTER(vp)], r2s_phys_reset_envs=[vp, vp, vp], r2s_phys_set_state_envs=[vp, vp, vp, vp, vp], r2s_phys_create_resting_case_envs=[vp, vp, vp], r2s_phys_mesh_motion=[vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
        r2s_phys_set_collision_lists=[vp, vp, vp, vp], r2s_phys_contact_stats=[vp, C.POINTER(C.c_int32), C.POINTER(vp)],
        r2s_phys_last_flavour=[vp, C.POINTER(C.c_int32)], r2s_phys_deferred_counts=[vp, vp, vp], r2s_phys_tagged_count=[vp, C.POINTER(C.c_int32), vp], r2s_phys_log_contacts=[vp, vp, vp], r2s_phys_set_tuning=[vp, i32, i32], r2s_phys_set_resident=[vp, i32], r2s_phys_set_pf=[vp, i32], r2s_phys_set_static_mesh_points=[vp, vp, C.c_int32, vp, vp], r2s_phys_side_stream=[i32, C.POINTER(vp)],
        r2s_phys_set_params=[vp, C.POINTER(R2SPhysParams), vp], r2s_phys_debug_pick_flavour=[C.POINTER(R2SFlavourIn), C.POINTER(R2SFlavourOut)], r2s_phys_debug_flavour_input=[vp, C.POINTER(R2SFlavourIn)], r2s_phys_last_flavour_ex=[vp, C.POINTER(R2SFlavourOut)], r2s_phys_check_fault=[vp, vp], r2s_phys_layout_stats=[vp, C.POINTER(C.c_int64)], r2s_phys_last_step_ms=[vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)],
    ).items():
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = args
    L.r2s_phys_set_timing.restype = None
    L.r2s_phys_set_timing.argtypes = [vp, i32]
    _bound = True
    return L


def _np(a, dtype):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def mesh_tuple(m):
    """Accept open3d-like meshes (``.vertices`` / ``.triangles``) or ``(vertices, triangles)`` tuples."""
    if hasattr(m, "vertices") and hasattr(m, "triangles"):
        return np.asarray(m.vertices, np.float32).reshape(-1, 3), np.asarray(m.triangles, np.int32).reshape(-1, 3)
    v, f = m
    return _np(v, np.float32).reshape(-1, 3), _np(f, np.int32).reshape(-1, 3)


class PhysBatch:
    """``n_env`` environments sharing one PhysTwin (springs, stiffness, masses, mesh topology).

    Parameter names follow ``SpringMassSystemWarp.__init__`` (sim/physics/spring_mass_warp.py:478-500);
    ``init_vertices`` is ``[n_env, N, 3]`` (or ``[N, 3]`` for one env)."""

    def __init__(self, *, init_vertices, init_springs, init_rest_lengths, init_masses, init_spring_Y, dt=5e-5,
                 num_substeps=667, dashpot_damping=100.0, drag_damping=3.0, spring_Y_min=0.0, spring_Y_max=1e5,
                 collision_dist=0.005, reverse_z=False, self_collision=True, collide_elas=0.5, collide_fric=0.3,
                 collide_eef_elas=0.0, collide_eef_fric=1.0, collide_self_elas=0.5, collide_self_fric=0.3,
                 init_collision_mask=None, init_velocities=None, dynamic_meshes=None, static_meshes=None,
                 use_pusher=False, collision_capacity=500, device="cuda:0"):
        L = _bind()
        self.device = torch.device(device)
        x = _np(init_vertices, np.float32)
        if x.ndim == 2:
            x = x[None]
        self.n_env, self.N = int(x.shape[0]), int(x.shape[1])
        springs = _np(init_springs, np.int32).reshape(-1, 2)
        self.S = int(springs.shape[0])
        rest = _np(init_rest_lengths, np.float32).reshape(-1)
        logy = _np(init_spring_Y, np.float32).reshape(-1)
        masses = _np(init_masses, np.float32).reshape(-1)[: self.N]
        vel = None if init_velocities is None else _np(init_velocities, np.float32).reshape(self.n_env, self.N, 3)
        masks = None if init_collision_mask is None else _np(init_collision_mask, np.int32).reshape(-1)[: self.N]
        if self_collision and masks is not None:
            assert np.unique(masks).shape[0] > 1  # spring_mass_warp.py:534
        dyn = [mesh_tuple(m) for m in (dynamic_meshes or [])]
        sta = [mesh_tuple(m) for m in (static_meshes or [])]
        meshes = dyn + sta
        self.num_substeps = int(num_substeps)
        self.use_pusher = bool(use_pusher)
        self.self_collision = bool(self_collision)
        f = lambda v: float(np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, dtype=np.float32).reshape(-1)[0])  # noqa: E731
        P = R2SPhysParams(f(dt), f(dashpot_damping), f(drag_damping), f(spring_Y_min), f(spring_Y_max), f(collision_dist),
                          f(collide_elas), f(collide_fric), f(collide_eef_elas), f(collide_eef_fric), f(collide_self_elas),
                          f(collide_self_fric), int(bool(reverse_z)), int(bool(self_collision)), int(bool(use_pusher)),
                          int(num_substeps))
        self.params = P
        d = R2SPhysDesc()
        d.params = P
        d.n_env, d.num_object_points, d.num_springs = self.n_env, self.N, self.S
        keep = [x, springs, rest, logy, masses, vel, masks]
        d.init_vertices = x.ctypes.data
        d.init_velocities = vel.ctypes.data if vel is not None else None
        d.init_springs = springs.ctypes.data if self.S else None
        d.init_rest_lengths = rest.ctypes.data if self.S else None
        d.init_spring_Y = logy.ctypes.data if self.S else None
        d.init_masses = masses.ctypes.data
        d.init_collision_mask = masks.ctypes.data if masks is not None else None
        d.n_dynamic_meshes, d.n_static_meshes = len(dyn), len(sta)
        self.n_dynamic_meshes = len(dyn)
        self.mesh_vertex_counts = np.array([len(v) for v, _ in meshes], np.int64)
        self.n_dyn_pts = int(sum(len(v) for v, _ in dyn))
        self.n_faces = int(sum(len(t) for _, t in meshes))
        if meshes:
            nv = np.array([len(v) for v, _ in meshes], np.int32)
            nf = np.array([len(t) for _, t in meshes], np.int32)
            vv = np.ascontiguousarray(np.concatenate([v for v, _ in meshes]).astype(np.float32))
            tt = np.ascontiguousarray(np.concatenate([t for _, t in meshes]).astype(np.int32))
            keep += [nv, nf, vv, tt]
            d.mesh_num_vertices, d.mesh_num_faces = nv.ctypes.data, nf.ctypes.data
            d.mesh_vertices, d.mesh_triangles = vv.ctypes.data, tt.ctypes.data
        d.collision_capacity = int(collision_capacity)
        self.collision_capacity = int(collision_capacity)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(L.r2s_phys_create(C.byref(d), C.byref(h), cur_stream(self.device)), "r2s_phys_create")
        self._h = h
        del keep
        # stable output tensors in the reference's layout; refreshed by sync_state()
        self.x = torch.empty(self.n_env, self.N, 3, dtype=torch.float32, device=self.device)
        self.v = torch.empty_like(self.x)
        self.sync_state()
        mm = np.zeros(self.n_faces, np.int32)
        fm = np.zeros(self.n_faces, np.int32)
        check(L.r2s_phys_mesh_maps(self._h, mm.ctypes.data, fm.ctypes.data), "r2s_phys_mesh_maps")
        self.mesh_map, self.face_map = mm, fm

    # -- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            _bind().r2s_phys_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------------
    def _s(self):
        return cur_stream(self.device)

    def sync_state(self):
        """Refresh ``self.x`` / ``self.v`` ([n_env, N, 3]) from the device state (wp.to_torch views in the reference)."""
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_get_state(self._h, self.x.data_ptr(), self.v.data_ptr(), self._s()), "r2s_phys_get_state")
        return self.x, self.v

    def set_state(self, x: torch.Tensor, v: Optional[torch.Tensor] = None):
        x = x.to(self.device, torch.float32).contiguous().reshape(self.n_env, self.N, 3)
        vp = None
        if v is not None:
            v = v.to(self.device, torch.float32).contiguous().reshape(self.n_env, self.N, 3)
            vp = v.data_ptr()
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_state(self._h, x.data_ptr(), vp, self._s()), "r2s_phys_set_state")
        self.sync_state()

    def set_state_envs(self, x: torch.Tensor, v: torch.Tensor, mask: torch.Tensor, resting_case: bool = False):
        """The particle state of an episode reset: rows of ``x`` / ``v`` ([n_env, N, 3]) of the environments with a non-zero ``mask``
        entry replace the device state, the others keep running untouched — including a pending fault of theirs, which the next
        ``step`` still reports (``set_state`` clears it; r2s_phys_set_state_envs).  ``resting_case``: also rebuild those
        environments' resting-pair set from the new positions (a reset into ANOTHER pose; the reference builds a new stepper)."""
        x = x.to(self.device, torch.float32).contiguous().reshape(self.n_env, self.N, 3)
        v = v.to(self.device, torch.float32).contiguous().reshape(self.n_env, self.N, 3)
        m = mask.to(self.device, torch.int32).contiguous().reshape(self.n_env)
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_state_envs(self._h, x.data_ptr(), v.data_ptr(), m.data_ptr(), self._s()), "r2s_phys_set_state_envs")
            if resting_case and self.self_collision:
                check(_bind().r2s_phys_create_resting_case_envs(self._h, m.data_ptr(), self._s()), "r2s_phys_create_resting_case_envs")
        self.sync_state()

    # -- per-env-step protocol (phystwin.py:362-521) ------------------------------------------------------
    def set_static_mesh_points(self, pts: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """Re-pose the static collision meshes of some environments (an episode reset into another scene pose; r2s_physics.h):
        ``pts`` float32 [n_env, n_static_vertices, 3] on the device, ``mask`` bool / int [n_env] or None = all."""
        pts = pts.to(self.device, torch.float32).contiguous()
        n_static = int((self.mesh_vertex_counts[self.n_dynamic_meshes:]).sum())
        assert tuple(pts.shape) == (self.n_env, n_static, 3), f"static mesh points must be [n_env, {n_static}, 3] (every static mesh, in order), got {tuple(pts.shape)}"
        m = None if mask is None else mask.to(self.device, torch.int32).contiguous()
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_static_mesh_points(self._h, pts.data_ptr(), n_static, 0 if m is None else m.data_ptr(), self._s()), "r2s_phys_set_static_mesh_points", reason=True)

    def create_resting_case(self):
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_create_resting_case(self._h, self._s()), "r2s_phys_create_resting_case")

    def update_collision_graph(self):
        assert self.self_collision
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_update_collision_graph(self._h, self._s()), "r2s_phys_update_collision_graph")

    def set_mesh_interactive(self, interp_points, interp_center, dynamic_velocity, dynamic_omega):
        E, n = self.n_env, self.num_substeps
        ndv = 1 if self.use_pusher else 2
        t = lambda a, shape: a.to(self.device, torch.float32).contiguous().reshape(shape)  # noqa: E731
        ip = t(interp_points, (E, n, self.n_dyn_pts, 3))
        ic = t(interp_center, (E, n, 3))
        dv = t(dynamic_velocity, (E, ndv, 3))
        om = t(dynamic_omega, (E, 1, 3))
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_mesh_interactive(self._h, ip.data_ptr(), ic.data_ptr(), dv.data_ptr(), om.data_ptr(), self._s()),
                  "r2s_phys_set_mesh_interactive")
        self._keep_mesh = (ip, ic, dv, om)

    # -- on-device gripper / pusher kinematics (the caller side of the stepper, phystwin.py:362-513) ----------------
    def set_eef_table(self, eef_pts_list, init_eef_xyz, grasp_force_threshold: float):
        """``eef_pts_list``: the knots of the reference's ``eef_pts_func`` (scipy interp1d over arange(K)/(K-1)),
        [K, n_dynamic_points, 3]; ``init_eef_xyz`` [3].  Resets current_openness / grasped of every environment."""
        tab = np.ascontiguousarray(np.asarray(eef_pts_list, np.float64))
        assert tab.ndim == 3 and tab.shape[1] == self.n_dyn_pts and tab.shape[2] == 3, tab.shape
        init = np.ascontiguousarray(np.asarray(init_eef_xyz, np.float32).reshape(-1)[:3])
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_eef_table(self._h, int(tab.shape[0]), tab.ctypes.data, init.ctypes.data, float(grasp_force_threshold), self._s()),
                  "r2s_phys_set_eef_table")

    def set_eef_motion(self, eef_xyz, eef_vel, eef_rot, eef_rot_vel, gripper_openness=None):
        """Per-environment inputs of ``SpringMassDynamicsModule.step``: eef_xyz/eef_vel/eef_rot_vel [n_env,3], eef_rot
        [n_env,3,3], gripper_openness [n_env] (device tensors).  Everything downstream happens on the device."""
        E = self.n_env
        t = lambda a, shape: a.to(self.device, torch.float32).contiguous().reshape(shape)  # noqa: E731
        x, v, r, w = t(eef_xyz, (E, 3)), t(eef_vel, (E, 3)), t(eef_rot, (E, 3, 3)), t(eef_rot_vel, (E, 3))
        o = t(gripper_openness, (E,)) if gripper_openness is not None else None
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_eef_motion(self._h, x.data_ptr(), v.data_ptr(), r.data_ptr(), w.data_ptr(),
                                                  o.data_ptr() if o is not None else None, self._s()), "r2s_phys_set_eef_motion")
        self._keep_eef = (x, v, r, w, o)

    def eef_state(self):
        """(current_openness float64 [n_env], grasped int32 [n_env]) copies."""
        from .raster import _memcpy_d2d

        po, pg = C.c_void_p(), C.c_void_p()
        check(_bind().r2s_phys_eef_state(self._h, C.byref(po), C.byref(pg)), "r2s_phys_eef_state")
        o = torch.empty(self.n_env, dtype=torch.float64, device=self.device)
        g = torch.empty(self.n_env, dtype=torch.int32, device=self.device)
        _memcpy_d2d(o.data_ptr(), po.value, o.numel() * 8, self.device)
        _memcpy_d2d(g.data_ptr(), pg.value, g.numel() * 4, self.device)
        return o.cpu(), g.cpu()

    def reset_envs(self, mask: Optional[torch.Tensor] = None):
        """Episode reset of the environments with a non-zero entry in ``mask`` ([n_env], default all): grasp state machine at its
        initial values, collision forces zero (r2s_phys_reset_envs).  Set their particle state with ``set_state``."""
        mp = None
        if mask is not None:
            mask = mask.to(self.device, torch.int32).contiguous().reshape(self.n_env)
            mp = mask.data_ptr()
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_reset_envs(self._h, mp, self._s()), "r2s_phys_reset_envs")

    def mesh_motion(self, points: bool = True):
        """Copies of the stepper's current motion inputs: interp_points [n_env, n_sub, n_dyn_pts, 3] (optional),
        interp_center [n_env, n_sub, 3], dynamic_velocity [n_env, 2, 3], dynamic_omega [n_env, 3]."""
        from .raster import _memcpy_d2d

        p = [C.c_void_p() for _ in range(4)]
        check(_bind().r2s_phys_mesh_motion(self._h, *[C.byref(q) for q in p]), "r2s_phys_mesh_motion")
        E, n = self.n_env, self.num_substeps
        shapes = [(E, n, self.n_dyn_pts, 3), (E, n, 3), (E, 2, 3), (E, 3)]
        out = []
        for k, (q, sh) in enumerate(zip(p, shapes)):
            if k == 0 and not points:
                out.append(None)
                continue
            t = torch.empty(*sh, dtype=torch.float32, device=self.device)
            _memcpy_d2d(t.data_ptr(), q.value, t.numel() * 4, self.device)
            out.append(t)
        return out

    def step(self, n_substeps: int = 0, first_substep: int = 0, sync_state: bool = True):
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_step(self._h, int(n_substeps), int(first_substep), self._s()), "r2s_phys_step", reason=True)
        if sync_state:
            self.sync_state()
        return self.x

    # -- read-backs -------------------------------------------------------------------------------------
    def collision_forces(self) -> torch.Tensor:
        """[n_env, n_faces, 3] copy of the last substep's per-face forces."""
        p, n = C.c_void_p(), C.c_int32()
        check(_bind().r2s_phys_collision_forces(self._h, C.byref(p), C.byref(n)), "r2s_phys_collision_forces")
        out = torch.empty(self.n_env, int(n.value), 3, dtype=torch.float32, device=self.device)
        if out.numel():
            from .raster import _memcpy_d2d
            _memcpy_d2d(out.data_ptr(), p.value, out.numel() * 4, self.device)
        return out

    def collision_lists(self):
        """(collision_number [n_env,N], collision_indices [n_env,N,cap]) copies."""
        num, idx, cap = C.c_void_p(), C.c_void_p(), C.c_int32()
        check(_bind().r2s_phys_collision_lists(self._h, C.byref(num), C.byref(idx), C.byref(cap)), "r2s_phys_collision_lists")
        from .raster import _memcpy_d2d
        tn = torch.empty(self.n_env, self.N, dtype=torch.int32, device=self.device)
        _memcpy_d2d(tn.data_ptr(), num.value, tn.numel() * 4, self.device)
        ti = None
        if idx.value:
            ti = torch.empty(self.n_env, self.N, int(cap.value), dtype=torch.int32, device=self.device)
            _memcpy_d2d(ti.data_ptr(), idx.value, ti.numel() * 4, self.device)
        return tn, ti

    def set_collision_lists(self, number, indices):
        """Write the candidate lists (the reference's collision_number / collision_indices arrays): ``number`` [n_env, N],
        ``indices`` [n_env, N, k] with k <= capacity, in the caller's particle indexing."""
        num = np.ascontiguousarray(_np(number, np.int32).reshape(self.n_env, self.N))
        idx = _np(indices, np.int32).reshape(self.n_env, self.N, -1)
        full = np.zeros((self.n_env, self.N, self.collision_capacity), np.int32)
        full[:, :, : idx.shape[2]] = idx
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_collision_lists(self._h, num.ctypes.data, full.ctypes.data, self._s()), "r2s_phys_set_collision_lists")

    def contact_stats(self):
        """(particles with self-collision candidates, mesh hits of the last substep summed over environments)."""
        from .raster import _memcpy_d2d

        n, p = C.c_int32(), C.c_void_p()
        check(_bind().r2s_phys_contact_stats(self._h, C.byref(n), C.byref(p)), "r2s_phys_contact_stats")
        hits = 0
        if p.value:
            t = torch.empty(self.n_env, dtype=torch.int32, device=self.device)
            _memcpy_d2d(t.data_ptr(), p.value, t.numel() * 4, self.device)
            hits = int(t.sum().item())
        return int(n.value), hits

    def log_contacts(self, out3: torch.Tensor):
        """{particles with candidates, mesh hits of the last substep, grasped envs} -> device int32[3], no host sync."""
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_log_contacts(self._h, out3.data_ptr(), self._s()), "r2s_phys_log_contacts")

    def deferred_counts(self) -> np.ndarray:
        """Deferred mesh queries per substep of the last env step (+ a trailing 'anything near a mesh' flag); diagnostics."""
        out = np.zeros(self.num_substeps + 1, np.int32)
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_deferred_counts(self._h, out.ctypes.data, self._s()), "r2s_phys_deferred_counts")
        return out

    def tagged_count(self) -> int:
        """Particles with self-collision candidates that were also handed to the finishing kernel's mesh list (tagged entries)
        at least once during the last ``step`` call; diagnostics."""
        n = C.c_int32()
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_tagged_count(self._h, C.byref(n), self._s()), "r2s_phys_tagged_count")
        return int(n.value)

    def last_flavour(self):
        """What the last ``step`` ran (r2s_phys_last_flavour_ex: the record of the pure flavour function, physics_flavour.h)."""
        o = R2SFlavourOut()
        check(_bind().r2s_phys_last_flavour_ex(self._h, C.byref(o)), "r2s_phys_last_flavour_ex")
        srv = int(o.n_srv)
        return dict(self_collision_kernel=bool(o.variant), mesh_template=int(o.mesh), deferred_mesh_queries=bool(o.mesh_defer) and not o.resident,
                    finishers_at_head_of_next_launch=bool(o.pf), chains=int(o.chains), resident=bool(o.resident), query_server_workgroups=srv,
                    servers_own_their_particle=bool(o.srv_own) and srv > 0, wavefronts_per_served_particle=(4 if o.srv_quad else 2) if srv else 0,
                    self_collision_servers=bool(o.self_srv), sum_class=int(o.sum_class), kernel=o.kernel.decode())

    def flavour_input(self):
        """The capabilities + switches of this handle as the flavour function reads them (a dict of R2SFlavourIn's fields)."""
        fin = R2SFlavourIn()
        check(_bind().r2s_phys_debug_flavour_input(self._h, C.byref(fin)), "r2s_phys_debug_flavour_input")
        return {n: int(getattr(fin, n)) for n, _ in R2SFlavourIn._fields_}

    def check_fault(self):
        """Raise NOW if a kernel of an earlier ``step`` flagged the state as invalid (a timed-out hand-off, an impulse beyond the bound a
        skipped mesh test relies on): waits for the stream.  ``step`` itself reports such a fault with a lag of up to two env steps, and a
        full ``set_state`` clears it — a caller that ends episodes and resets everything at once asks here first (r2s_phys_check_fault)."""
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_check_fault(self._h, self._s()), "r2s_phys_check_fault", reason=True)

    def set_tuning(self, chains: int = 0, mesh_defer: int = -1):
        check(_bind().r2s_phys_set_tuning(self._h, int(chains), int(mesh_defer)), "r2s_phys_set_tuning")

    def side_stream(self, k: int = 1):
        """The device's pooled side stream k as a torch stream (r2s_physics.h: one of the streams the env step's kernel chains run on;
        idle between env steps)."""
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_side_stream(int(k), C.byref(p)), "r2s_phys_side_stream")
        return torch.cuda.ExternalStream(p.value, device=self.device)

    def set_resident(self, on: bool):
        """Small batches only (r2s_physics.h): run the env step's free flavour as one resident launch (default) or, off, with
        the per-substep kernels of the same 64-particle layout."""
        check(_bind().r2s_phys_set_resident(self._h, int(bool(on))), "r2s_phys_set_resident")

    def set_pf(self, on: bool):
        """Large batches only (r2s_physics.h): the contact flavours with the finishers of substep k at the head of substep k + 1's launch
        (default) or, off, as two launches per substep.  Bit-identical states."""
        check(_bind().r2s_phys_set_pf(self._h, int(bool(on))), "r2s_phys_set_pf")

    def collision_max_count(self) -> int:
        m = C.c_int32()
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_collision_max_count(self._h, C.byref(m), self._s()), "r2s_phys_collision_max_count")
        return int(m.value)

    def set_spring_Y(self, log_Y):
        a = _np(log_Y, np.float32).reshape(-1)
        assert a.shape[0] == self.S
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_spring_Y(self._h, a.ctypes.data, self._s()), "r2s_phys_set_spring_Y")

    _FLOAT_PARAMS = ("dt", "dashpot_damping", "drag_damping", "spring_Y_min", "spring_Y_max", "collision_dist", "collide_elas", "collide_fric",
                     "collide_eef_elas", "collide_eef_fric", "collide_self_elas", "collide_self_fric")

    def set_params(self, **kw):
        """set_collide / set_collide_object and friends (spring_mass_warp.py:955-995): scalar parameters only; the structural
        fields (num_substeps, self_collision, use_pusher, reverse_z) are fixed at construction."""
        for k, v in kw.items():
            if k not in self._FLOAT_PARAMS:
                raise ValueError(f"{k!r} is not a settable scalar parameter (structural fields are fixed at construction)")
            setattr(self.params, k, float(np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, np.float32).reshape(-1)[0]))
        with torch.cuda.device(self.device):
            check(_bind().r2s_phys_set_params(self._h, C.byref(self.params), self._s()), "r2s_phys_set_params")

    def layout_stats(self):
        a = (C.c_int64 * 8)()
        check(_bind().r2s_phys_layout_stats(self._h, a), "r2s_phys_layout_stats")
        k = ["blocks", "halo_max", "ell_slots", "neighbour_slots", "fallback_slots", "lds_bytes", "chains", "blocks_per_xcd"]
        return dict(zip(k, [int(v) for v in a]))

    def set_timing(self, on: bool):
        _bind().r2s_phys_set_timing(self._h, int(on))

    def last_step_ms(self):
        ms, k = C.c_float(), C.c_int32()
        check(_bind().r2s_phys_last_step_ms(self._h, C.byref(ms), C.byref(k)), "r2s_phys_last_step_ms")
        return float(ms.value), int(k.value)
