"""Scenario builders shared by the physics tests: the same inputs go to the CPU oracle (oracle.PhysOracle)
and to the HIP path (sim.physics.SpringMassSystemWarp / r2s_hip.physics.PhysBatch)."""
from types import SimpleNamespace

import numpy as np

DEFAULTS = dict(dt=5e-5, num_substeps=667, dashpot_damping=100.0, drag_damping=3.0, spring_Y_min=0.0, spring_Y_max=1e5,
                collision_dist=0.005, reverse_z=False, self_collision=True, collide_elas=0.5, collide_fric=0.3,
                collide_eef_elas=0.0, collide_eef_fric=1.0, collide_self_elas=0.5, collide_self_fric=0.3)


def cfg(**over):
    """A stand-in for the hydra ``phystwin_cfg`` node (cfg/physics/default.yaml)."""
    d = dict(DEFAULTS, init_spring_Y=3e4, use_graph=True, collision_requires_grad=False)
    d.update(over)
    return SimpleNamespace(**d)


def make_object(shape="rope", n=600, seed=0, lift=0.0):
    from r2s_hip import synth

    ob = synth.phystwin_object(shape, n, seed)
    ob["points"] = ob["points"].copy()
    ob["points"][:, 2] += lift
    return ob


def two_blobs(seed=0, gap=0.06, speed=3.0, n=150):
    """Two separate soft blobs; the second flies towards the first along +x (self-collision scenario)."""
    from r2s_hip import synth

    a = synth.lattice_points("sloth", n, seed)
    b = synth.lattice_points("sloth", n, seed + 1)
    a[:, 2] += 0.05
    b[:, 2] += 0.05
    b[:, 0] += (a[:, 0].max() - b[:, 0].min()) + gap
    pts = np.concatenate([a, b]).astype(np.float32)
    sa, ra = synth.build_springs(a)
    sb, rb = synth.build_springs(b)
    springs = np.concatenate([sa, sb + len(a)]).astype(np.int32)
    rest = np.concatenate([ra, rb]).astype(np.float32)
    rng = np.random.default_rng(seed)
    logy = np.log(rng.uniform(5e3, 3e4, len(springs))).astype(np.float32)
    v = np.zeros_like(pts)
    v[len(a):, 0] = -speed
    return dict(points=pts, springs=springs, rest=rest, log_Y=logy, v0=v)


def two_sheets(seed=0, n=144, gap=0.004, z0=0.03):
    """Two single-layer sheets standing upright in the x-z plane, `gap` apart in y (closer than collision_dist = 5 mm): every
    particle of one sheet is a self-collision candidate of the sheet opposite once the candidate lists are rebuilt, AND within
    the 5 mm margin of a finger pad pressed on the sheet — the two contact kinds on the SAME particles.  Returns the object
    (points in the close configuration) and the number of particles of the first sheet; construct the steppers from
    `far_apart(ob, nA)` (the resting-pair set must not contain the cross-sheet pairs), then set the state to ob["points"]."""
    from r2s_hip import synth

    a = synth.lattice_points("cloth", n, seed)
    b = synth.lattice_points("cloth", n, seed + 1)

    def up(p, y):
        return np.stack([p[:, 0], np.full(len(p), y) + (p[:, 2] - p[:, 2].mean()), p[:, 1] - p[:, 1].min() + z0], 1).astype(np.float32)

    A, B = up(a, -gap / 2), up(b, +gap / 2)
    sa, ra = synth.build_springs(A)
    sb, rb = synth.build_springs(B)
    pts = np.concatenate([A, B])
    springs = np.concatenate([sa, sb + len(A)]).astype(np.int32)
    rest = np.concatenate([ra, rb]).astype(np.float32)
    rng = np.random.default_rng(seed)
    logy = np.log(rng.uniform(5e3, 3e4, len(springs))).astype(np.float32)
    return dict(points=pts, springs=springs, rest=rest, log_Y=logy, v0=np.zeros_like(pts)), len(A)


def far_apart(ob, nA, shift=0.3):
    far = dict(ob)
    far["points"] = ob["points"].copy()
    far["points"][nA:, 0] += shift
    return far


def oracle_env(ob, f64=False, dynamic_meshes=None, static_meshes=None, use_pusher=False, **over):
    import oracle

    kw = dict(DEFAULTS)
    kw.update(over)
    return oracle.PhysOracle(ob["points"], ob["springs"], ob["rest"], ob["log_Y"], v0=ob.get("v0"), f64=f64,
                             dynamic_meshes=dynamic_meshes, static_meshes=static_meshes, use_pusher=use_pusher, **kw)


def hip_env(ob, device="cuda:0", dynamic_meshes=None, static_meshes=None, use_pusher=False, n_env=1, **over):
    import torch
    from r2s_hip.physics import PhysBatch

    kw = dict(DEFAULTS)
    kw.update(over)
    x = np.repeat(ob["points"][None], n_env, 0)
    v = None if ob.get("v0") is None else np.repeat(ob["v0"][None], n_env, 0)
    return PhysBatch(init_vertices=x, init_springs=ob["springs"], init_rest_lengths=ob["rest"],
                     init_masses=np.ones(len(ob["points"]), np.float32), init_spring_Y=ob["log_Y"], init_velocities=v,
                     dynamic_meshes=dynamic_meshes, static_meshes=static_meshes, use_pusher=use_pusher, device=device, **kw)


def gripper_motion(fingers, n_sub, dt, vel=(0.0, 0.0, -0.4), omega=(0.0, 0.0, 0.0), closing=0.3):
    """Rigid finger motion over one env step in the reference's parametrisation (phystwin.py:374-452):
    interpolated vertices per substep, eef centre per substep, per-finger velocity, angular velocity."""
    pts0 = np.concatenate([v for v, _ in fingers]).astype(np.float64)
    center0 = pts0.mean(0)
    nl = len(fingers[0][0])
    ts = (np.arange(1, n_sub + 1) * dt)[:, None, None]
    vel = np.asarray(vel, np.float64)
    close_dir = np.zeros_like(pts0)
    close_dir[:nl, 1] = +closing  # left finger moves +y, right finger -y
    close_dir[nl:, 1] = -closing
    interp = pts0[None] + vel[None, None] * ts + close_dir[None] * ts
    centers = center0[None] + vel[None] * ts[:, 0]
    dyn_vel = np.stack([vel * 0.5 + np.array([0, closing * 0.5, 0]), vel * 0.5 - np.array([0, closing * 0.5, 0])])
    dyn_omega = -np.asarray(omega, np.float64)[None] * 0.5
    return interp.astype(np.float32), centers.astype(np.float32), dyn_vel.astype(np.float32), dyn_omega.astype(np.float32)


def rigid_motion(mesh, n_sub, dt, vel=(0.1, 0.0, 0.0), omega=(0.0, 0.0, 0.0)):
    """One rigid dynamic mesh (the pusher) over one env step in the reference's parametrisation (phystwin.py:462-510):
    per-substep vertices, centre, dynamic_velocity [1,3] (= half the eef velocity), dynamic_omega [1,3] (= -half the rate)."""
    pts0 = np.asarray(mesh[0], np.float64)
    c0 = pts0.mean(0)
    vel, omega = np.asarray(vel, np.float64), np.asarray(omega, np.float64)
    interp = np.empty((n_sub, len(pts0), 3))
    centers = np.empty((n_sub, 3))
    for s in range(n_sub):
        t = (s + 1) * dt
        ang = np.linalg.norm(omega) * t
        if ang > 0:
            k = omega / np.linalg.norm(omega)
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        else:
            R = np.eye(3)
        centers[s] = c0 + vel * t
        interp[s] = (pts0 - c0) @ R.T + centers[s]
    return interp.astype(np.float32), centers.astype(np.float32), (vel * 0.5)[None].astype(np.float32), (-omega * 0.5)[None].astype(np.float32)
