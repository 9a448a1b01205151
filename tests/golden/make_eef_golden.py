"""Generates tests/golden/eef_step.npz by EXECUTING THE REFERENCE's own ``SpringMassDynamicsModule.step``
(/root/reference/sim/physics/phystwin.py:362-521) on the CPU: the gripper branch (:367-460 — openness / grasp state machine on
the summed finger-face forces, finger-vertex interpolation, per-substep rigid motion, halved eef velocity + closing velocity,
negated half rate) and the pusher branch (:462-513).  The module is imported from where it lies, not copied:

* ``warp`` is the float32 interpreter shim tests/golden/warp_shim.py (phystwin.py imports warp and spring_mass_warp at module top;
  ``step`` itself only calls ``wp.capture_launch`` / ``simulator.step`` at its end — the recorder simulator below makes that a no-op);
* ``open3d``, ``sapien``, ``transforms3d``, ``urdfpy`` are empty placeholder modules (imported by the module, unused by ``step``);
* ``kornia`` is a placeholder carrying ONE function, ``axis_angle_to_rotation_matrix`` — third party, absent from the image, restated
  below from kornia 0.7's published source; that conversion alone is not the reference's code here ("parity unpinned" for it);
* the module object is created without ``__init__`` (which builds the warp simulator from a PhysTwin checkpoint) and given the
  attributes ``step`` reads: ``phystwin_cfg`` (dt, num_substeps, self_collision, grasp_force_threshold, use_graph), ``device``,
  ``use_pusher``, ``current_openness``, ``grasped`` and a RECORDER ``simulator`` with ``mesh_map.numpy()``, scripted
  ``collision_forces.numpy()`` and a ``set_mesh_interactive`` that captures its four arguments;
* ``eef_pts_func`` is built exactly like robot_pc_transformations.py:190 / :225 builds it (scipy ``interp1d`` over 101 openings),
  on the synthetic finger / rod vertex table stored in the fixture.

Scripted sequences (all inputs are stored next to the outputs):
  gripper A (40 substeps): open -> closing -> both finger forces above grasp_force_threshold 3e4 (grasp: opening held) -> held ->
      forces between 100 and the threshold while still commanded closed (the 0.05-per-step creep) -> one finger only above the
      threshold (no new grasp) -> both below 100 (release; the opening follows the command again) -> re-opening;
      with translation, rotation rate (incl. the first-order branch of the axis-angle conversion) and a rotated frame;
  gripper B (667 substeps, the real count): three steps, vertices stored at substeps 0, 1, 333, 665, 666 only;
  pusher (40 substeps): a 62-vertex rod, three steps.

Usage (authoring container only):  python tests/golden/make_eef_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import scipy.interpolate
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path[:0] = [HERE, os.path.join(ROOT, "real2sim-eval_amd")]


def kornia_axis_angle_to_rotation_matrix(axis_angle):
    """kornia.geometry.conversions.axis_angle_to_rotation_matrix (kornia 0.7), restated: (N, 3) -> (N, 3, 3)."""
    def _rodrigues(aa, theta2, eps=1e-6):
        theta = torch.sqrt(theta2)
        wxyz = aa / (theta + eps)
        wx, wy, wz = torch.chunk(wxyz, 3, dim=1)
        c, s = torch.cos(theta), torch.sin(theta)
        r00 = c + wx * wx * (1.0 - c)
        r10 = wz * s + wx * wy * (1.0 - c)
        r20 = -wy * s + wx * wz * (1.0 - c)
        r01 = wx * wy * (1.0 - c) - wz * s
        r11 = c + wy * wy * (1.0 - c)
        r21 = wx * s + wy * wz * (1.0 - c)
        r02 = wy * s + wx * wz * (1.0 - c)
        r12 = -wx * s + wy * wz * (1.0 - c)
        r22 = c + wz * wz * (1.0 - c)
        return torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)

    def _taylor(aa):
        rx, ry, rz = torch.chunk(aa, 3, dim=1)
        k = torch.ones_like(rx)
        return torch.cat([k, -rz, ry, rz, k, -rx, -ry, rx, k], dim=1).view(-1, 3, 3)

    _aa = torch.unsqueeze(axis_angle, dim=1)
    theta2 = torch.matmul(_aa, _aa.transpose(1, 2)).squeeze(1)
    normal, taylor = _rodrigues(axis_angle, theta2), _taylor(axis_angle)
    mask = (theta2 > 1e-6).view(-1, 1, 1)
    mask_pos = mask.type_as(theta2)
    mask_neg = (~mask).type_as(theta2)
    out = torch.eye(3).to(axis_angle.device).type_as(axis_angle).view(1, 3, 3).repeat(axis_angle.shape[0], 1, 1)
    out[..., :3, :3] = mask_pos * normal + mask_neg * taylor
    return out


def load_reference():
    import warp_shim

    wp = warp_shim.warp
    wp.init = lambda: None
    wp.ScopedTimer = types.SimpleNamespace(enabled=False)
    wp.set_module_options = lambda *a, **k: None
    wp.capture_launch = lambda g: None
    wp.to_torch = lambda a: a                      # step() returns current_points = wp.to_torch(simulator.wp_state.wp_x) (:521-526)
    sys.modules["warp"] = wp
    for name in ("open3d", "sapien", "sapien.core", "transforms3d", "urdfpy", "kornia", "kornia.geometry", "kornia.geometry.conversions"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["urdfpy"].URDF = object
    sys.modules["sapien"].core = sys.modules["sapien.core"]
    k = sys.modules["kornia"]
    k.geometry = sys.modules["kornia.geometry"]
    k.geometry.conversions = sys.modules["kornia.geometry.conversions"]
    k.geometry.conversions.axis_angle_to_rotation_matrix = kornia_axis_angle_to_rotation_matrix
    sys.path.insert(0, "/root/reference")
    return importlib.import_module("sim.physics.phystwin")


class _Arr:
    def __init__(self, a):
        self.a = a

    def numpy(self):
        return self.a


class RecorderSimulator:
    """Stands where SpringMassSystemWarp stands for ``step``: mesh_map / collision_forces readbacks, set_mesh_interactive capture."""

    def __init__(self, mesh_map, n_faces):
        self.mesh_map = _Arr(np.asarray(mesh_map, np.int32))
        self.collision_forces = _Arr(np.zeros((n_faces, 3), np.float32))
        self.graph = None
        self.wp_state = types.SimpleNamespace(wp_x=torch.zeros(1, 3), wp_v=torch.zeros(1, 3))
        self.calls = []
        self.rebuilds = 0

    def update_collision_graph(self):
        self.rebuilds += 1

    def set_mesh_interactive(self, pts, center, vel, omega):
        self.calls.append([np.array(t.detach().cpu().numpy(), np.float32) for t in (pts, center, vel, omega)])

    def step(self):
        pass


def make_module(P, mesh_map, n_faces, n_sub, use_pusher, dt=5e-5, thr=3e4):
    m = object.__new__(P.SpringMassDynamicsModule)
    m.phystwin_cfg = types.SimpleNamespace(dt=dt, num_substeps=n_sub, self_collision=False, grasp_force_threshold=thr, use_graph=False)
    m.device = "cpu"
    m.use_pusher = use_pusher
    m.simulator = RecorderSimulator(mesh_map, n_faces)
    m.current_openness = None
    m.grasped = False
    return m


def run(P, m, fn, init, steps, keep=None):
    """steps: list of dicts xyz, vel, rot, rv, open, force.  Returns stacked inputs / outputs."""
    out = {k: [] for k in ("xyz", "vel", "rot", "rv", "open", "force", "pts", "center", "dvel", "omega", "cur", "grasped")}
    for s in steps:
        if s.get("force") is not None:
            m.simulator.collision_forces.a = np.asarray(s["force"], np.float32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))  # noqa: E731
        op = None if m.use_pusher else torch.tensor([s["open"]], dtype=torch.float32)
        m.step(t(s["xyz"]), t(s["vel"]), t(s["rot"]), t(s["rv"]), op, fn, torch.from_numpy(np.asarray(init, np.float32)))
        pts, center, dvel, omega = m.simulator.calls[-1]
        out["pts"].append(pts if keep is None else pts[keep])
        out["center"].append(center if keep is None else center[keep])
        out["dvel"].append(dvel); out["omega"].append(omega)
        out["cur"].append(np.float64(m.current_openness)); out["grasped"].append(bool(m.grasped))
        for k in ("xyz", "vel", "rot", "rv"):
            out[k].append(np.asarray(s[k], np.float32))
        out["open"].append(np.float32(1.0 if m.use_pusher else s["open"]))
        out["force"].append(m.simulator.collision_forces.a.copy())
    return {k: np.stack(v) for k, v in out.items()}


def finger_forces(mesh_map, left, right):
    """Per-face forces whose filtered sums (faces 18 + 19 + 1 of each finger, phystwin.py:390-391) have norms ``left`` / ``right``."""
    f = np.zeros((len(mesh_map), 3), np.float32)
    l0, r0 = np.flatnonzero(mesh_map == 0)[0], np.flatnonzero(mesh_map == 1)[0]
    f[l0 + 18] = (0.6 * left, 0, 0); f[l0 + 19] = (0, 0, 0.8 * left); f[l0 + 1] = (0.4 * left, 0, 0); f[l0 + 18, 0] -= 0.4 * left
    f[r0 + 18] = (0, 0.5 * right, 0); f[r0 + 1] = (0, 0.5 * right, 0)
    f[l0 + 5] = (9e5, 9e5, 0); f[r0 + 30] = (0, -9e5, 9e5)              # faces the filter must ignore
    return f


def main():
    from scipy.spatial.transform import Rotation
    from r2s_hip import synth

    P = load_reference()
    rng = np.random.default_rng(2024)
    tab, init, fl, fr = synth.gripper_eef_table()
    fn = scipy.interpolate.interp1d(np.arange(101) / 100.0, tab, axis=0)                 # robot_pc_transformations.py:190
    mesh_map = np.concatenate([np.zeros(len(fl), np.int32), np.ones(len(fr), np.int32), -np.ones(12, np.int32)])   # + a static box
    nF = len(mesh_map)
    THR = 3e4

    # ---- gripper A: the whole state machine, 40 substeps -------------------------------------------------------------------
    #           command  left    right     what the reference's step must decide
    script = [(1.00, 0.0, 0.0),            # first call: current_openness := command
              (0.80, 0.0, 0.0),            # closing, no force: follows
              (0.55, 50.0, 20.0),          # closing, forces < 100: follows (grasped stays False)
              (0.40, 3.5e4, 3.2e4),        # closing, both > 3e4: GRASP, opening held at 0.55
              (0.20, 4.0e4, 5.0e4),        # still both large: held at 0.55
              (0.20, 2.0e4, 3.5e4),        # one below the threshold, grasped: creep to 0.50
              (0.20, 500.0, 900.0),        # creep to 0.45
              (0.43, 500.0, 900.0),        # creep bounded by the command: max(0.43, 0.40) = 0.43
              (0.43, 500.0, 900.0),        # command == current: the else branch (follows), still grasped
              (0.30, 99.0, 3.5e4),         # one finger < 100 only: not released; one > thr only: no new hold -> creep 0.38
              (0.30, 99.0, 99.9),          # both < 100: RELEASED, follows the command (0.30)
              (0.10, 150.0, 3.1e4),        # not grasped, one large: follows
              (0.60, 3.5e4, 3.5e4),        # opening with large forces: follows (the test is only taken while closing)
              (1.30, 0.0, 0.0),            # command above 1: current_openness 1.3, vertices at clip(1.3) = 1
              (-0.2, 0.0, 0.0)]            # below 0: clipped to 0 for the vertices
    stepsA = []
    xyz = np.array([[0.41, 0.03, 0.32]], np.float32)
    rot = Rotation.from_euler("xyz", [0.15, -0.1, 0.5]).as_matrix().astype(np.float32)[None]
    for k, (cmd, lf, rf) in enumerate(script):
        vel = rng.uniform(-0.1, 0.1, (1, 3)).astype(np.float32)
        rv = rng.uniform(-1.0, 1.0, (1, 3)).astype(np.float32)
        if k in (1, 8):
            rv = (rv * 1e-1).astype(np.float32)     # theta^2 of the last substeps around 1e-6: both branches of the conversion inside one step
        if k == 2:
            rv[:] = 0
        stepsA.append(dict(xyz=xyz.copy(), vel=vel, rot=rot.copy(), rv=rv, open=cmd, force=finger_forces(mesh_map, lf, rf)))
        xyz = (xyz + vel * np.float32(40 * 5e-5)).astype(np.float32)
        rot = (Rotation.from_rotvec(rv[0].astype(np.float64) * 40 * 5e-5).as_matrix().T @ rot[0].astype(np.float64)).astype(np.float32)[None]
    mA = make_module(P, mesh_map, nF, 40, False, thr=THR)
    A = run(P, mA, fn, init, stepsA)

    # ---- gripper B: the real substep count ------------------------------------------------------------------------------------
    keep = np.array([0, 1, 333, 665, 666])
    stepsB = []
    xyz = np.array([[0.35, -0.02, 0.28]], np.float32)
    rot = Rotation.from_euler("xyz", [np.pi, 0.0, 0.3]).as_matrix().astype(np.float32)[None]
    for cmd, lf, rf, vz in [(1.0, 0, 0, -0.09), (0.5, 0, 0, -0.02), (0.2, 3.3e4, 3.4e4, 0.05)]:
        vel = np.array([[0.01, -0.02, vz]], np.float32)
        rv = np.array([[0.02, 0.3, -0.4]], np.float32)
        stepsB.append(dict(xyz=xyz.copy(), vel=vel, rot=rot.copy(), rv=rv, open=cmd, force=finger_forces(mesh_map, lf, rf)))
        xyz = (xyz + vel * np.float32(667 * 5e-5)).astype(np.float32)
    mB = make_module(P, mesh_map, nF, 667, False, thr=THR)
    B = run(P, mB, fn, init, stepsB, keep=keep)

    # ---- pusher ----------------------------------------------------------------------------------------------------------------
    rod_v, rod_f = synth.cylinder_mesh((0.0, 0.0, -0.1), radius=0.005, length=0.2, n_seg=10, n_rings=5)   # 62 vertices, 120 faces
    initP = np.array([0.3, 0.0, 0.4], np.float32)
    rel = rod_v.astype(np.float64).copy(); rel[:, 1] *= -1; rel[:, 2] *= -1
    tabP = np.repeat((initP.astype(np.float64) + rel)[None], 101, axis=0)
    fnP = scipy.interpolate.interp1d(np.arange(101) / 100.0, tabP, axis=0)             # robot_pc_transformations.py:225
    mapP = np.zeros(len(rod_f), np.int32)
    stepsP = []
    xyz = np.array([[0.2, 0.1, 0.21]], np.float32)
    rot = Rotation.from_euler("xyz", [0.0, 0.05, -0.3]).as_matrix().astype(np.float32)[None]
    for k in range(3):
        vel = rng.uniform(-0.1, 0.1, (1, 3)).astype(np.float32)
        rv = rng.uniform(-0.5, 0.5, (1, 3)).astype(np.float32)
        stepsP.append(dict(xyz=xyz.copy(), vel=vel, rot=rot.copy(), rv=rv, open=1.0, force=None))
        xyz = (xyz + vel * np.float32(40 * 5e-5)).astype(np.float32)
    mP = make_module(P, mapP, len(rod_f), 40, True, thr=THR)
    Pz = run(P, mP, fnP, initP, stepsP)

    out = dict(dt=5e-5, thr=THR, mesh_map=mesh_map, table=tab, init_eef_xyz=init, faces_left=fl, faces_right=fr,
               B_keep=keep, P_table=tabP, P_init_eef_xyz=initP, P_mesh_map=mapP, P_faces=rod_f)
    for tag, d, n in (("A", A, 40), ("B", B, 667), ("P", Pz, 40)):
        out[f"{tag}_n_sub"] = n
        for k, v in d.items():
            out[f"{tag}_{k}"] = v
    path = os.path.join(HERE, "eef_step.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
    print("A current_openness:", A["cur"].tolist())
    print("A grasped:", A["grasped"].tolist())
    print("B current_openness:", B["cur"].tolist(), B["grasped"].tolist())
    print("P dvel shape", Pz["dvel"].shape, "pts", Pz["pts"].shape)


if __name__ == "__main__":
    main()
