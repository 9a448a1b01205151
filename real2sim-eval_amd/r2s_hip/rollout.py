"""Thin batched rollout driver: the loop the reference's ``BaseEnv.step`` + ``get_obs`` run per environment
(sim/envs/env.py:53-94 -> PhysTwinDynamics.step, phystwin.py:362-521 -> GSRenderer.render / render_wrist,
gs_renderer.py:924-1000), for ``n_env`` environments of one GPU at once.

One env step (``BatchedRollout.step(action)``; the reference: ``env.step({'action': (1, 13)})``, env.py:86-94) =
    action -> end-effector motion per environment (phystwin.py:104-147: velocity and angular rate from the commanded next pose)
    update_collision_graph                       (once per env step, phystwin.py:365-366)
    gripper / pusher kinematics + grasp logic    (on device: r2s_phys_set_eef_motion; phystwin.py:367-513)
    num_substeps fused physics substeps          (the captured graph, phystwin.py:515-519)
    Gaussians follow their particles             (LBS skinning, incremental: gs_renderer.py:717-747 -> r2s_skin_*)
    wrist camera of every environment from its NEW end-effector pose (on device: r2s_wrist_camera; gs_renderer.py:966-985)
    2 rasterised frames per env (side + wrist)   (env.py:55-56)

Synthetic inputs only: there is no network for the real PhysTwin / Scaniverse assets.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import synth
from .physics import PhysBatch
from .raster import RasterBatch
from .skinning import Skinning, knn_relations, knn_weights

CONFIGS = {
    # name: (object shape, particles, gaussians per env, envs, W, H)  — BASELINE.json configs[1..3]
    "rope_1env": ("rope", 8000, 40000, 1, 640, 480),
    # the same rope folded onto itself (rest shape a hairpin, the upper leg sags onto the lower one): live self-collision candidates
    # through the window, the gripper hovering — the resident stepper's self-collision flavour
    "rope_fold_1env": ("rope_fold", 8000, 40000, 1, 640, 480),
    "rope_tip_fold_1env": ("rope_tip_fold", 2000, 10000, 1, 320, 240),   # a test scene: 8 cm of the rope's end folded back onto it
    "sloth_32env": ("sloth_arms", 15000, 80000, 32, 640, 480),   # configs[2]: the headline workload
    "T_32env": ("T", 2229, 40000, 32, 640, 480),
    "T_pusher_32env": ("T", 2229, 40000, 32, 640, 480),   # configs[3] per GPU: T block pushed by the ~25k-face pusher rod
    "sloth_multicam_8env": ("sloth_arms", 15000, 140000, 8, 1280, 720),   # configs[4] per GPU: 4 views, +60k robot-link Gaussians
    "tiny": ("rope", 600, 3000, 2, 160, 120),
}


def axis_angle_to_rotation_matrix(aa: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.conversions.axis_angle_to_rotation_matrix restated (kornia is not a dependency here): [N,3] -> [N,3,3],
    Rodrigues with aa / (theta + 1e-6), first-order matrix when theta^2 <= 1e-6 — the conversion phystwin.py:379 applies."""
    theta2 = (aa * aa).sum(1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = aa / (theta + 1e-6)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s_ = torch.cos(theta), torch.sin(theta)
    k = 1.0 - c
    normal = torch.cat([c + wx * wx * k, wx * wy * k - wz * s_, wy * s_ + wx * wz * k, wz * s_ + wx * wy * k, c + wy * wy * k, -wx * s_ + wy * wz * k,
                        -wy * s_ + wx * wz * k, wx * s_ + wy * wz * k, c + wz * wz * k], dim=1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    return torch.where((theta2 > 1e-6).view(-1, 1, 1), normal, taylor)


def rotation_matrix_to_axis_angle(R: torch.Tensor) -> torch.Tensor:
    """Log map of a batch of rotation matrices [N,3,3] -> [N,3] (the role of kornia's rotation_matrix_to_axis_angle at
    phystwin.py:137; kornia goes through a quaternion, this is the same map written directly — parity unpinned, kornia absent)."""
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    cos = ((tr - 1.0) * 0.5).clamp(-1.0, 1.0)
    theta = torch.acos(cos)
    v = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1)   # 2 sin(theta) axis
    s_ = torch.sin(theta)
    scale = torch.where(theta > 1e-4, theta / (2.0 * s_.clamp(min=1e-12)), torch.full_like(theta, 0.5))
    return v * scale[:, None]


def rotation_matrix_to_quaternion(R: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """[N,3,3] -> [N,4] (w, x, y, z): the role of kornia.geometry.conversions.rotation_matrix_to_quaternion at phystwin.py:117 /
    gs_renderer's eef_quat (kornia >= 0.7: scalar first).  The published branch scheme (largest of trace / diagonal entries picks
    the component computed by the square root, sq = 2 sqrt(. + eps)); the sign is the one that scheme yields — w > 0 on the trace
    branch, the pivot component > 0 otherwise — not canonicalised, like kornia's.  Parity unpinned (kornia absent): pinned against
    scipy up to the sign in tests/test_host_logic.py."""
    m = R.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[:, k] for k in range(9)]
    tr = m00 + m11 + m22

    s0 = torch.sqrt(tr + 1.0 + eps) * 2.0
    d0 = s0.clamp(min=torch.finfo(R.dtype).tiny)
    q0 = torch.stack([0.25 * s0, (m21 - m12) / d0, (m02 - m20) / d0, (m10 - m01) / d0], 1)
    s1 = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
    d1 = s1.clamp(min=torch.finfo(R.dtype).tiny)
    q1 = torch.stack([(m21 - m12) / d1, 0.25 * s1, (m01 + m10) / d1, (m02 + m20) / d1], 1)
    s2 = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
    d2 = s2.clamp(min=torch.finfo(R.dtype).tiny)
    q2 = torch.stack([(m02 - m20) / d2, (m01 + m10) / d2, 0.25 * s2, (m12 + m21) / d2], 1)
    s3 = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
    d3 = s3.clamp(min=torch.finfo(R.dtype).tiny)
    q3 = torch.stack([(m10 - m01) / d3, (m02 + m20) / d3, (m12 + m21) / d3, 0.25 * s3], 1)
    w2 = torch.where((m11 > m22)[:, None], q2, q3)
    w1 = torch.where(((m00 > m11) & (m00 > m22))[:, None], q1, w2)
    return torch.where((tr > 0.0)[:, None], q0, w1)


def quaternion_to_rotation_matrix(q: torch.Tensor) -> torch.Tensor:
    """[N,4] (w, x, y, z) -> [N,3,3], the quaternion normalised first (kornia.geometry.conversions.quaternion_to_rotation_matrix at
    phystwin.py:111 and experiments/eval_policy.py: the policy's eef_quat -> the action's rotation block)."""
    q = torch.nn.functional.normalize(q.reshape(-1, 4), dim=1, eps=1e-12)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    return torch.stack([1.0 - ty * y - tz * z, tx * y - tz * w, tx * z + ty * w,
                        tx * y + tz * w, 1.0 - tx * x - tz * z, ty * z - tx * w,
                        tx * z - ty * w, ty * z + tx * w, 1.0 - tx * x - ty * y], 1).reshape(-1, 3, 3)


def env_shifts(n_env, seed):
    """Per-environment planar shift of the object (a few cm; environment 0 unshifted): the stand-in for the grid randomisation of
    cfg/gs/*.yaml when a rollout is built without episode ids."""
    rng = np.random.default_rng(seed + 1)
    sh = np.zeros((n_env, 3), np.float32)
    sh[:, :2] = rng.uniform(-0.03, 0.03, (n_env, 2))
    sh[0] = 0
    return sh


def scene_setup(config, seed=0, num_substeps=667, schedule=None, close_at=15, open_at=10**9, with_gripper=True, with_static=True,
                close_rate=None, lift_steps=None):
    """The synthetic scene of a config as plain numpy: the PhysTwin object, the collision meshes (two finger meshes or
    the pusher rod, the box obstacle), the end effector's table of finger vertices and start pose, and the parameters of the synthetic
    action trace (``eef_velocity`` / ``open_command``).  No device is touched: ``BatchedRollout`` builds its batch from it, and the
    checker's closed loop (oracle/closed_loop.py) the same scene on the CPU.

    ``close_rate``: how fast the commanded opening falls once the gripper closes, per env step.  None (the class default, what every
    scenario test of rounds 1-5 was written against): the command JUMPS to its closed value in env step ``close_at`` — a gripper that
    snaps shut within 1/30 s, and one that can never detect a grasp: the reference only establishes a grasp while the command is still
    BELOW the current opening (phystwin.py:399-405), and after a jump the two are equal from the next step on.  A rate (bench.py and
    evaluate use 0.1: the stroke from open to the closed value in ~7 env steps, like a gripper driven by a policy at 30 Hz) ramps the
    command down, the pads load up step by step, and the state machine runs as it does in an episode: closing -> grasped (both pads'
    filtered forces > 3e4) -> opening frozen / creeping 0.05 per step -> held through the lift."""
    shape, n_particles = CONFIGS[config][0], CONFIGS[config][1]
    ob = synth.phystwin_object(shape, n_particles, seed)
    pts = ob["points"]
    c, top = pts.mean(0), pts[:, 2].max()
    ob_shape = "sloth" if shape == "sloth_arms" else ("rope" if shape.startswith("rope") else shape)
    use_pusher = "pusher" in config
    scn = dict(config=config, ob=ob, ob_shape=ob_shape, use_pusher=use_pusher, schedule="push", close_at=0, open_at=10**9, v_down=0.1,
               eef_table=None, eef_init=None, eef0=None, dyn=[], sta=[], num_substeps=int(num_substeps), dt=5e-5,
               close_rate=None if close_rate is None else float(close_rate), faces_left=None, faces_right=None,
               lift_steps=None if lift_steps is None else int(lift_steps))
    # commanded opening while closed.  With a closing rate: 0 — "close", as a policy commands it; where the fingers stop is the grasp
    # state machine's business (phystwin.py:399-405: the opening freezes once both pads' forces exceed the threshold).  Without one
    # (the command jumps): 0.3 (34 mm between the pads) squeezes the toy's 61 mm pair of arms; the 24 mm rope needs 0.08 (18 mm): with
    # 0.3 the pads stop at the rope's 5 mm contact margin, brush it and lift nothing
    scn["closed_cmd"] = 0.0 if close_rate is not None else (0.08 if ob_shape == "rope" else 0.3)
    if with_gripper and use_pusher:
        # vertical pusher rod next to the block's -x face (assets/.../pusher_20cm.stl has 25 368 faces)
        scn["eef_init"] = np.array([0.3, 0.0, 0.4], np.float32)
        rod_v, rod_f = synth.cylinder_mesh((0.0, 0.0, -0.1), radius=0.005, length=0.2)
        rel = rod_v.astype(np.float64)
        rel[:, 1] *= -1
        rel[:, 2] *= -1
        scn["eef_table"] = np.repeat((scn["eef_init"].astype(np.float64) + rel)[None], 2, axis=0)   # rigid: two equal knots
        # the rod (radius 5 mm, 1 mm contact margin) pushes along +x at 5 cm/s and reaches the block's -x face at env step `close_at`
        scn["schedule"], scn["close_at"] = "push", int(close_at)
        ys = np.sort(pts[pts[:, 0] < pts[:, 0].min() + 0.01, 1])
        y_face = float(ys[(len(ys) - 1) // 2])   # the T's bar: the part that reaches furthest in -x (the LOWER median: torch.nanmedian's, which a posed reset uses)
        scn["eef0"] = np.array([pts[:, 0].min() - 0.006 - 0.05 * scn["close_at"] * num_substeps * 5e-5, y_face, 0.2], np.float32)
        scn["rod_offset_x"] = float(scn["eef0"][0] - pts[:, 0].min())   # (negative: the rod starts on the -x side) — a posed reset keeps this distance to the turned block
        scn["dyn"] = [(synth.eef_world_points(scn["eef_table"][-1], scn["eef_init"], scn["eef0"]), rod_f)]
    elif with_gripper:
        scn["eef_table"], scn["eef_init"], fl, fr = synth.gripper_eef_table()
        scn["faces_left"], scn["faces_right"] = fl, fr
        scn["schedule"] = schedule or ("grasp" if shape == "sloth_arms" or config == "rope_1env" else "lissajous")
        scn["close_at"], scn["open_at"] = (int(close_at), int(open_at)) if scn["schedule"] == "grasp" else (100, 300)
        if scn["schedule"] == "grasp":
            # fingers (5 cm tall, centred 6 cm below the eef) end up centred on the arms' upper 8 cm: eef = top + 2 cm at
            # `close_at`, reached by a straight descent at 0.1 m/s (3.3 mm per env step); for a flat object (rope, T block) the
            # finger tips stop 2 mm above the table instead of going through it
            z_close = max(top + 0.02, 0.002 + 0.025 + 0.06)
            scn["eef0"] = np.array([c[0], c[1], z_close + scn["v_down"] * scn["close_at"] * num_substeps * 5e-5], np.float32)
        else:
            scn["eef0"] = np.array([c[0], c[1], top + 0.1], np.float32)
        w0 = synth.eef_world_points(scn["eef_table"][-1], scn["eef_init"], scn["eef0"])
        scn["dyn"] = [(w0[: len(w0) // 2], fl), (w0[len(w0) // 2:], fr)]
    if with_static:
        scn["sta"] = [synth.box_mesh((c[0] + 0.25, c[1] + 0.2, 0.135), (0.2, 0.13, 0.27))]
    return scn


def eef_velocity(scn, step):
    """End-effector velocity of env step ``step`` of the scene's synthetic action trace (m/s, float32 [3])."""
    w = 2 * np.pi * 0.25
    tt = step / 30.0
    if scn["use_pusher"]:  # push along +x at 5 cm/s with a slow lateral weave
        return np.array([0.05, 0.02 * np.cos(w * tt), 0.0], np.float32)
    if scn["schedule"] == "grasp":
        if step < scn["close_at"]:      # free motion: straight down over the arms, fingers open
            return np.array([0.0, 0.0, -scn["v_down"]], np.float32)
        if step < scn["close_at"] + close_steps(scn):     # the fingers close during these env steps (one, without a closing rate)
            return np.zeros(3, np.float32)
        # lift, with a slow sway along the fingers; after `lift_steps` env steps (None: never) the end effector keeps its height (a full
        # episode is 450+ steps: an unbounded lift would carry the object out of every camera's view)
        lift = scn.get("lift_steps")
        up = 0.05 if lift is None or step < scn["close_at"] + close_steps(scn) + lift else 0.0
        return np.array([0.02 * np.cos(w * tt), 0.0, up], np.float32)
    return np.array([0.05 * w * np.cos(w * tt) * 0.6, 0.05 * w * np.cos(2 * w * tt + 0.5) * 0.6, -0.01 * np.sin(w * tt)], np.float32)


def close_steps(scn):
    """Env steps the commanded opening takes from 1 to the scene's closed value."""
    if scn["close_rate"] is None:
        return 1
    return int(np.ceil((1.0 - scn["closed_cmd"]) / scn["close_rate"] - 1e-9))


def open_command(scn, step):
    """Commanded gripper opening (1 = open) of env step ``step``: closed between ``close_at`` and ``open_at`` — at once, or ramped down
    by ``close_rate`` per env step (``scene_setup``)."""
    if not (scn["close_at"] <= step < scn["open_at"]):
        return 1.0
    if scn["close_rate"] is None:
        return float(scn["closed_cmd"])
    return float(max(scn["closed_cmd"], 1.0 - scn["close_rate"] * (step - scn["close_at"] + 1)))


class BatchedRollout:
    def __init__(self, config="sloth_32env", device="cuda:0", seed=0, n_env=None, num_substeps=667, views=2,
                 self_collision=True, with_gripper=True, with_static=True, tile_culling=True, schedule=None, close_at=15, open_at=10**9,
                 settle_steps=None, randomize=False, res=None, close_rate=None, lift_steps=None):
        """``schedule``: the synthetic action trace.  "grasp" (default for the sloth scenes and rope_1env): the open gripper comes down over
        the object (free motion) — the toy's raised arms, the middle of the rope —, closes on it at env step
        ``close_at`` — finger contact, for the toy the two arms pressed together (live self-collision candidates), grasp detection —
        and lifts: the reference's episodes are spent with the gripper ON the object (eval_policy.py:124-213), so a timed window has a
        contact half.  "lissajous" (the default of the small test scenes and of T_32env — a gripper over the push-T block is
        not a scene of the reference, which pushes it with the rod: T_pusher_32env): the gripper hovers 10 cm above the object on a Lissajous path, closes at
        step 100, opens at 300 — no contact inside a short window.  The pusher scene pushes along +x and reaches the block at
        ``close_at``.  ``close_rate`` / ``lift_steps``: see ``scene_setup`` — None (the default here) JUMPS the commanded opening to its
        closed value within env step ``close_at`` (every scenario test of rounds 1-5 was written against that; such a gripper can never
        detect a grasp); bench.py and the grasp tests pass a rate, the opening ramps down and the grasp state machine runs as in an episode.
        ``res``: (W, H) instead of the config's frame size (the reference's default frame is 848x480,
        cfg/env/xarm_gripper.yaml:21-49).  ``settle_steps`` (default 40 for "grasp"): env steps run
        inside the constructor with the gripper parked 15 cm higher, so that the toy is AT REST when the rollout starts
        (SURVEY.md §8d places the objects resting on the table; a jittered lattice with random stiffness is not in equilibrium
        under gravity and its arms sway by ~1 cm for the first second).  ``randomize``: ``reset(env_ids, episode_ids=...)`` places
        each reset environment's object at the grid pose of its episode id (``episode_pose``; the reference: env.reset(seed=episode_id)
        -> load_scaniverse(randomize, index), gs_renderer.py:333-637) — costs one rotation array per environment."""
        shape, n_particles, n_gauss, envs, W, H = CONFIGS[config]
        if res is not None:
            W, H = int(res[0]), int(res[1])
        self.config = config
        self.n_env = int(n_env if n_env is not None else envs)
        if "multicam" in config:
            views = 4
        self.W, self.H, self.views = W, H, views
        self.device = torch.device(device)
        self.num_substeps = int(num_substeps)
        self.dt = 5e-5
        E = self.n_env
        # the scene itself — object, collision meshes, the end effector's start pose and action trace — is plain numpy and shared with the
        # checker's closed loop (oracle/closed_loop.py builds the same scene without a GPU)
        scn = self.scene = scene_setup(config, seed=seed, num_substeps=num_substeps, schedule=schedule, close_at=close_at, open_at=open_at,
                                       with_gripper=with_gripper, with_static=with_static, close_rate=close_rate, lift_steps=lift_steps)
        ob = self.ob = scn["ob"]
        self.ob_shape = scn["ob_shape"]
        self.schedule, self.close_at, self.open_at = scn["schedule"], scn["close_at"], scn["open_at"]
        pts = ob["points"]
        self.N, self.S = len(pts), len(ob["springs"])
        # per-env pose: grid randomisation stand-in — planar shifts of a few cm (cfg/gs/*.yaml patterns)
        self.env_shift = env_shifts(E, seed)
        x0 = pts[None] + self.env_shift[:, None]
        c = pts.mean(0)
        top = pts[:, 2].max()
        self.use_pusher = scn["use_pusher"]
        self.eef_table, self.eef_init, self.eef0 = scn["eef_table"], scn["eef_init"], scn["eef0"]
        self.v_down = scn["v_down"]
        dyn, sta = scn["dyn"], scn["sta"]
        self.fingers = dyn
        self.phys = PhysBatch(init_vertices=x0, init_springs=ob["springs"], init_rest_lengths=ob["rest"],
                              init_masses=np.ones(self.N, np.float32), init_spring_Y=ob["log_Y"], num_substeps=num_substeps,
                              self_collision=self_collision, dynamic_meshes=dyn, static_meshes=sta, use_pusher=self.use_pusher,
                              collide_eef_fric=0.2 if self.use_pusher else 1.0, device=self.device)
        self.with_gripper = with_gripper
        self._springs_dev = torch.from_numpy(np.ascontiguousarray(ob["springs"], np.int32)).to(self.device)
        self._target_dev = torch.from_numpy(np.ascontiguousarray(pts + np.array([0.10, 0.0, 0.0], np.float32))).to(self.device)  # push-T goal: 10 cm along +x
        self._box = (np.array([c[0] + 0.25, c[1] + 0.2, 0.135]), 0.5 * np.array([0.2, 0.13, 0.27]))
        # per-environment pose of the box obstacle (an episode reset with `randomize` may re-pose it: episode_mesh_poses)
        # vertices of EVERY static mesh, in the constructor's order (r2s_phys_set_static_mesh_points takes them all), and which mesh each belongs to
        self._static_v0 = torch.from_numpy(np.ascontiguousarray(np.concatenate([v for v, _ in sta]), np.float32)).to(self.device) if sta else None
        self._static_mesh_of = torch.from_numpy(np.concatenate([np.full(len(v), k, np.int64) for k, (v, _) in enumerate(sta)])).to(self.device) if sta else None
        self._static_c0 = torch.tensor([float(c[0] + 0.25), float(c[1] + 0.2), 0.0], device=self.device)   # its origin: turned about, in the table plane
        self._box_c = torch.tensor(self._box[0], dtype=torch.float32, device=self.device)[None].repeat(E, 1)
        self._box_R = torch.eye(3, device=self.device)[None].repeat(E, 1, 1)
        self._box_posed = False
        # Gaussians: object splats ride on particles, table splats are static; configs[4] adds 60k splats on the robot's links
        self.with_robot = "multicam" in config
        n_scene = n_gauss - 60000 if self.with_robot else n_gauss
        sc = synth.gaussian_scene(n_scene, seed, object_points=pts)
        n_tab = int(n_scene * 0.35)
        self.n_obj = n_scene - n_tab
        if self.with_robot:
            self.arm_offsets = synth.arm_offsets(seed)
            self.arm_base = synth.arm_fk(np.deg2rad(synth.ARM_INIT_QPOS_DEG), finger=0.05)
            rs, rlink = synth.robot_scan(n_gauss - n_scene, seed, self.arm_offsets, self.arm_base)
            self.scan_mask = np.concatenate([np.full(n_tab, -1, np.int32), rlink])   # total_mask of the table + robot scan
            sc = {k: np.concatenate([sc[k], rs[k]]) for k in sc}
        self.P = len(sc["means3D"])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        # LBS topology, once per scene (gs_renderer.py:195-211): 8-NN among bones (= particles), 16 nearest bones per Gaussian
        obj0 = sc["means3D"][: self.n_obj]
        w, wi = knn_weights(pts, obj0, min(16, self.N))
        self.skin = Skinning(knn_relations(pts, min(8, self.N - 1)), w, wi, device=self.device)
        self.means = torch.empty(E, self.P, 3, dtype=torch.float32, device=self.device)
        self.means[:, : self.n_obj] = t(obj0)[None] + t(self.env_shift)[:, None]
        self.means[:, self.n_obj:] = t(sc["means3D"][self.n_obj:])[None]
        self.t = 0
        self.bones = self.phys.x.clone()                          # particle positions the Gaussians currently correspond to
        self.g = {k: t(v) for k, v in sc.items() if k != "means3D"}
        self.rot_env = None
        self.randomize = bool(randomize)
        self.random_variables = {}                                # episode id -> [x, y, z, angle] as the reference records them (gs_renderer.py:637)
        self.random_mesh_variables = {}                           # episode id -> [[x, y, z, angle] per static mesh with a grid] (gs_renderer.py:390)
        if self.randomize and not self.with_robot:
            self.rot_env = torch.nn.functional.normalize(self.g["rotations"], dim=-1)[None].repeat(E, 1, 1).contiguous()
        if self.with_robot:
            # per-environment rotations (the robot's splats turn with their links); object / table rows are the shared values
            from .robot import RobotGaussians
            self.rot_env = torch.nn.functional.normalize(self.g["rotations"], dim=-1)[None].repeat(E, 1, 1).contiguous()
            self.robot = RobotGaussians(synth.ARM_LINKS, synth.ARM_LISTED, self.arm_offsets, self.arm_base, sc["means3D"][self.n_obj:],
                                        sc["rotations"][self.n_obj:], self.scan_mask, device=self.device)
            self._robot_first = True
        self.raster = RasterBatch(self.device)
        self.raster.set_tile_culling(tile_culling)  # exact-output instance culling (include/r2s_raster.h)
        # Sync-free batches: no host read of the instance count inside the pipeline after the first batch.  A batch whose count
        # outgrew the capacity derived from the previous one (+12.5 %: a reset, an object entering the view) loses its deepest
        # instances; `render` polls without blocking and counts such batches (`lossy_batches`), `observations()` — what a
        # closed-loop caller reads — waits, checks, and re-renders the step synchronously if it was one of them.
        self.raster.set_async(True)
        self._raster_overflows, self.lossy_batches = 0, 0
        self.cams = [synth.side_camera(W, H), synth.wrist_camera(W, H, eef_pos=(c[0], c[1], top + 0.30)),
                     synth.orbit_camera(W, H, 60.0, target=c), synth.orbit_camera(W, H, -110.0, target=c)][:views]
        self.cam_t = [{k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()} for cam in self.cams]
        # The wrist camera (view 1) rides on each environment's end effector: its matrices are rebuilt on the device every step
        # from (eef_xyz, eef_rot) and the fixed calibration eef2c (gs_renderer.py:966-985), one row per environment, written
        # into the arrays the prepared frames point at.  The synthetic gripper table (synth.gripper_eef_table) is expressed in a
        # tool frame whose y and z axes are flipped with respect to the xarm's (the physics inputs use eef_rot = identity for a
        # tool pointing down), so the calibration constant absorbs the flip: eef2c = inv(c2eef) . diag(1, -1, -1, 1).
        self.wrist = None
        if views >= 2:
            from .camera import WristCamera
            flip = np.diag([1.0, -1.0, -1.0, 1.0])
            self.wrist = WristCamera(E, W, H, synth.scaled_K(synth.WRIST_K, W, H), np.linalg.inv(synth.WRIST_C2EEF) @ flip, device=self.device)
            self._wrist_xyz = t(np.array([c[0], c[1], top + 0.30], np.float32))[None].repeat(E, 1) + t(self.env_shift)   # without a gripper: parked
            self._wrist_rot = torch.eye(3, device=self.device).repeat(E, 1, 1)
        self.out_color = torch.empty(E, views, 3, H, W, dtype=torch.float32, device=self.device)
        self.out_depth = torch.empty(E, views, 1, H, W, dtype=torch.float32, device=self.device)
        self._update_means()
        self._sets = [RasterBatch.make_set(self.means[e], self.g["opacities"], shs=self.g["shs"], scales=self.g["scales"],
                                           rotations=self.rot_env[e] if self.rot_env is not None else self.g["rotations"]) for e in range(E)]
        self._frames = []
        for e in range(E):
            for vi, cam in enumerate(self.cam_t):
                vm, pm, cp = cam["viewmatrix"], cam["projmatrix"], cam["campos"]
                if vi == 1 and self.wrist is not None:          # per environment, rewritten in place every step
                    vm, pm, cp = self.wrist.viewmatrix[e], self.wrist.projmatrix[e], self.wrist.campos[e]
                self._frames.append(dict(set=e, viewmatrix=vm, projmatrix=pm, campos=cp,
                                         bg=cam["bg"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], z_threshold=cam["z_threshold"],
                                         out_color=self.out_color[e, vi], out_depth=self.out_depth[e, vi]))
        self.t = 0
        self._vel_trace = None
        self._dephase, self._env_delay, self._restarted = 1, None, False
        self._cand_fresh = False
        self._prepared = None
        self._log = None
        self.last_num_rendered = 0
        if with_gripper:
            self._init_gripper_motion()
        n_settle = (40 if self.schedule == "grasp" else 0) if settle_steps is None else int(settle_steps)
        if n_settle > 0 and with_gripper:
            park = torch.tensor([0.0, 0.0, 0.15], device=self.device)
            zero = torch.zeros(E, 3, device=self.device)
            for _ in range(n_settle):
                if self.phys.self_collision:
                    self.phys.update_collision_graph()
                self.phys.set_eef_motion(self.eef_xyz + park, zero, self.eef_rot, self.eef_rot_vel, None if self.use_pusher else torch.ones(E, device=self.device))
                self.phys.step(0, 0, sync_state=False)
            self.phys.set_eef_table(self.eef_table, self.eef_init, 3e4)   # forget the parked pose: current_openness = None, grasped = False
            self.phys.sync_state()
            self._update_means()
        # what an episode starts from (reset): particles at rest, the object's Gaussians on them, the end effector at its start pose
        self._env_t0 = torch.zeros(E, dtype=torch.long, device=self.device)
        self._init = dict(x=self.phys.x.clone(), v=self.phys.v.clone(), means=self.means[:, : self.n_obj].clone())
        if self.rot_env is not None:
            self._init["rot"] = self.rot_env[:, : self.n_obj].clone()
        if with_gripper:
            self._init.update(eef_xyz=self.eef_xyz.clone(), eef_rot=self.eef_rot.clone())
        self._obj_center = torch.tensor([float(c[0]), float(c[1]), 0.0], device=self.device)
        # nothing has been rendered yet: the first get_obs() / observations() renders the start state (BaseEnv.reset returns get_obs())
        self._frame_dirty = True

    # ---- end-effector trace: fixed Lissajous path at <= 0.1 m/s; the gripper closes at step 100 and opens at 300 ------
    # (SURVEY.md §8d).  Only the eef pose / rates / commanded opening are produced here — what BaseEnv hands to
    # SpringMassDynamicsModule.step (phystwin.py:362); finger vertices, grasp logic and per-substep motion are on device.
    def _init_gripper_motion(self):
        E = self.n_env
        self.phys.set_eef_table(self.eef_table, self.eef_init, 3e4)   # cfg/physics/default.yaml:51 grasp_force_threshold
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        self.eef_xyz = t(np.repeat(self.eef0[None], E, 0) + self.env_shift)
        self.eef_rot = torch.eye(3, device=self.device).repeat(E, 1, 1)
        self.eef_rot_vel = torch.zeros(E, 3, device=self.device)
        self.eef_gripper = torch.ones(E, device=self.device)      # commanded opening of the last step (state['eef_gripper'], phystwin.py:165)

    def _eef_velocity(self, step):
        return eef_velocity(self.scene, step)

    def synthetic_action(self, step):
        """The synthetic action trace as a CALLER of ``step``: the end-effector motion of env step ``step`` for every environment,
        device tensors (the whole trace lives on the device: no per-step upload).  ``dephase`` (set by ``set_dephase``) delays
        environment e by ``(e * dephase) // n_env`` steps along the same trace, so that the environments are not all in the same phase of
        their episode (episodes of eval_policy_parallel.py are independent: they do not share a phase)."""
        E = self.n_env
        if self._vel_trace is None or step >= len(self._vel_trace) - self._dephase:   # (a restarted episode only looks further back)
            n = max(1024, 2 * (step + 1 + self._dephase))
            self._vel_trace = torch.from_numpy(np.stack([self._eef_velocity(k) for k in range(n)])).to(self.device)
            self._open_cmd = torch.tensor([open_command(self.scene, k) for k in range(n)], dtype=torch.float32, device=self.device)
        if self._dephase > 1 or self._restarted:
            idx = step - self._env_t0                             # an environment's episode starts at its last reset ...
            if self._dephase > 1:
                idx = idx - self._env_delay                       # ... and environment e replays the trace (e * dephase) // n_env steps late
            act, idc = idx >= 0, idx.clamp(min=0)
            vel = torch.where(act[:, None], self._vel_trace[idc], torch.zeros(E, 3, device=self.device))
            openness = torch.where(act, self._open_cmd[idc], torch.ones(E, device=self.device))
        else:
            vel = self._vel_trace[step][None].expand(E, 3).contiguous()
            openness = self._open_cmd[step].expand(E).contiguous()
        return dict(eef_xyz=self.eef_xyz, eef_vel=vel, eef_rot=self.eef_rot, eef_rot_vel=self.eef_rot_vel,
                    gripper_openness=None if self.use_pusher else openness, eef_rot_next=self.eef_rot)   # the traces do not rotate

    def set_dephase(self, k: int):
        """Stagger the synthetic episodes over a window of ``k`` env steps: environment e starts its trace ``(e * k) // n_env`` steps late
        (it waits, gripper open, at the start pose) — the delays cover 0 .. k - 1 evenly whatever the number of environments, so over
        the window the same share of env-steps is in contact as when all environments close at its middle.  k <= 1: all in phase."""
        self._dephase = max(1, int(k))
        self._env_delay = (torch.arange(self.n_env, device=self.device) * self._dephase) // self.n_env
        self._vel_trace = None

    def action13_to_motion(self, action, fps=None):
        """The reference's 'xyz_rot' action (phystwin.py:113-118, :131-138): ``action`` [n_env, 13] = next end-effector position
        (3), next rotation matrix (9, row-major), commanded opening (1) -> the motion of this env step, all on the device:
        eef_vel = (xyz_next - xyz) fps, eef_rot_vel = axis_angle(eef_rot . inv(rot_next)) fps."""
        a = action.to(self.device, torch.float32).reshape(self.n_env, 13)
        fps = float(fps) if fps is not None else 1.0 / (self.num_substeps * self.dt)
        xyz_next, rot_next = a[:, :3], a[:, 3:12].reshape(-1, 3, 3)
        vel = (xyz_next - self.eef_xyz) * fps
        delta = self.eef_rot.bmm(torch.linalg.inv(rot_next))
        return dict(eef_xyz=self.eef_xyz, eef_vel=vel, eef_rot=self.eef_rot, eef_rot_vel=rotation_matrix_to_axis_angle(delta) * fps,
                    gripper_openness=None if self.use_pusher else a[:, 12].contiguous(), eef_xyz_next=xyz_next, eef_rot_next=rot_next)

    def apply_action(self, action):
        """Hand one env step's end-effector motion to the stepper — what BaseEnv hands to SpringMassDynamicsModule.step
        (phystwin.py:362) — for every environment: a dict of device tensors eef_xyz [E,3], eef_vel [E,3], eef_rot [E,3,3],
        eef_rot_vel [E,3], gripper_openness [E] (None for the pusher), or an [E,13] 'xyz_rot' action tensor.  Advances the
        rollout's end-effector pose (what the wrist camera and the next action start from) like phystwin.py:160-166."""
        if torch.is_tensor(action):
            action = self.action13_to_motion(action)
        E = self.n_env
        xyz, vel = action["eef_xyz"].reshape(E, 3), action["eef_vel"].reshape(E, 3)
        rot, rv = action["eef_rot"].reshape(E, 3, 3), action["eef_rot_vel"].reshape(E, 3)
        self.phys.set_eef_motion(xyz, vel, rot, rv, action.get("gripper_openness"))
        if action.get("gripper_openness") is not None:
            self.eef_gripper = action["gripper_openness"].reshape(E)
        T = self.num_substeps * self.dt
        self.eef_xyz = action["eef_xyz_next"] if "eef_xyz_next" in action else xyz + vel * T
        if "eef_rot_next" in action:
            self.eef_rot = action["eef_rot_next"]
        else:   # integrate the commanded rate like the stepper does (phystwin.py:377-380 at the last substep); device ops, no sync
            self.eef_rot = axis_angle_to_rotation_matrix(rv * T).transpose(1, 2).bmm(rot)

    def _arm_qpos(self, step):
        """Synthetic joint trajectory per environment (stand-in for the policy's qpos): slow sinusoids about the init pose."""
        base = np.deg2rad(synth.ARM_INIT_QPOS_DEG)
        ph = np.arange(self.n_env)[:, None] * 0.7 + np.arange(7)[None] * 1.3
        q = base[None] + 0.25 * np.sin(0.08 * step + ph)
        finger = 0.05 * (0.5 + 0.5 * np.cos(0.05 * step + np.arange(self.n_env)))
        return q, finger

    def _update_means(self):
        """Scene assembly for the rasteriser, in place in its per-environment Gaussian sets (no torch.cat, no copies):
        * object splats: incremental skinning as update_rendervar does it — bones = particles at the last render, motions = what
          they moved since, applied to the last skinned Gaussian positions (gs_renderer.py:738-747, :762-769, :1096);
        * robot splats (configs[4]): rigidly with their links (robot_pc_transformations.py:12-55), FK poses as input."""
        x = self.phys.x
        obj = self.means[:, : self.n_obj]
        self.skin.interpolate_motions(self.bones, x - self.bones, obj, out=obj)
        self.bones.copy_(x)
        if self.with_robot:
            q, finger = self._arm_qpos(self.t)
            pose = np.stack([synth.arm_fk(q[e], finger[e]) for e in range(self.n_env)])
            k = self._pose_k = (getattr(self, "_pose_k", -1) + 1) % 64      # pinned ring: the upload never blocks the host
            if not hasattr(self, "_pose_pin"):
                self._pose_pin = torch.empty(64, *pose.shape, dtype=torch.float32).pin_memory()
                self._pose_dev = torch.empty(64, *pose.shape, dtype=torch.float32, device=self.device)
            self._pose_pin[k].copy_(torch.from_numpy(pose))
            self._pose_dev[k].copy_(self._pose_pin[k], non_blocking=True)
            self.link_pose = self._pose_dev[k]
            self.robot.transform(self.link_pose, self.means[:, self.n_obj:], self.rot_env[:, self.n_obj:], normalize=True, write_static=self._robot_first)
            self._robot_first = False

    # ---- episode reset of some environments ------------------------------------------------------------------------
    # grid randomisation of the object pose per scene: cfg/gs/{rope,sloth,T}.yaml `object.grid_randomization` (xy in m, theta in degrees)
    # `meshes`: the grid of every static mesh that has one (cfg/gs/sloth.yaml `meshes[0].grid_randomization`: the box obstacle; the rope's clip
    # and the T scene have none), in the order of the config
    GRIDS = {"rope": dict(xy=[(-0.05, -0.05), (-0.05, 0.0), (-0.05, 0.05), (0.0, -0.05), (0.0, 0.0), (0.0, 0.05), (0.05, -0.05), (0.05, 0.0), (0.05, 0.05)],
                          theta=[-10, 0, 10], one_to_one=False, meshes=[]),
             "sloth": dict(xy=[(0, 0), (-0.05, 0), (0.05, 0), (0, -0.05), (0, 0.03)], theta=[0, -5, 5, -5, 5], one_to_one=True,
                           meshes=[dict(xy=[(0, 0), (-0.05, 0), (0.05, 0), (0, 0.05)], theta=[0, -5, 5, 5], one_to_one=True)]),
             "T": dict(xy=[(-0.05, -0.05), (-0.05, 0.05), (0.05, -0.05), (0.05, 0.05)], theta=[45, 135, 225, 315], one_to_one=False, meshes=[])}

    @staticmethod
    def _grid_entry(g, i):
        if g["one_to_one"]:
            xy, th = g["xy"][i], g["theta"][i]
        else:
            xy, th = g["xy"][i // len(g["theta"])], g["theta"][i % len(g["theta"])]
        return float(xy[0]), float(xy[1]), 0.0, float(th) * np.pi / 180.0

    @staticmethod
    def _grid_size(g):
        return len(g["xy"]) if g["one_to_one"] else len(g["xy"]) * len(g["theta"])

    def episode_pose(self, episode_id: int):
        """(x, y, z = 0, angle in radians) of the OBJECT in episode ``episode_id``: the reference's grid arithmetic (gs_renderer.py:340-348,
        :600-637: true_index = episode_id mod n_object_rand; one_to_one: xy[i], theta[i]; else xy[i // n_theta], theta[i % n_theta])."""
        g = self.GRIDS[self.ob_shape]
        return self._grid_entry(g, int(episode_id) % self._grid_size(g))

    def episode_mesh_poses(self, episode_id: int):
        """[(x, y, z = 0, angle)] of every static mesh with a grid in episode ``episode_id``: what is left of the index above the object's
        grid is peeled mesh by mesh (gs_renderer.py:347, :365-383: true_index_mesh = index // n_object_rand; per mesh: this = true_index_mesh
        mod n_this_mesh, true_index_mesh //= n_this_mesh) — the sloth scene: 5 object poses x 4 box poses = 20 scenes."""
        g = self.GRIDS[self.ob_shape]
        rest = int(episode_id) // self._grid_size(g)
        out = []
        for m in g["meshes"]:
            n = self._grid_size(m)
            out.append(self._grid_entry(m, rest % n))
            rest //= n
        return out

    def _pose_static_meshes(self, mesh_pose, mask):
        """Re-pose the static collision meshes of the environments in ``mask`` (device bool [n_env]): ``mesh_pose`` [n_env, G, 4] = (x, y, z, angle)
        of the first G static meshes (those with a grid; the rest keep their loaded pose).  A mesh turns about the scene's mesh origin and
        shifts (gs_renderer.py:385-388: pose[:3, 3] += t, pose[:3, :3] = Rz pose[:3, :3]); the stepper gets ALL static vertices of those
        environments (r2s_phys_set_static_mesh_points), the success predicate's box follows mesh 0."""
        E, dev = self.n_env, self.device
        G = mesh_pose.shape[1]
        per_v = mesh_pose[:, self._static_mesh_of.clamp(max=G - 1)]                               # [E, V, 4]: every vertex' mesh pose ...
        per_v = torch.where((self._static_mesh_of < G)[None, :, None], per_v, torch.zeros_like(per_v))   # ... identity for meshes without a grid
        ca, sa = torch.cos(per_v[..., 3]), torch.sin(per_v[..., 3])
        d = self._static_v0[None] - self._static_c0
        v_new = torch.stack([ca * d[..., 0] - sa * d[..., 1], sa * d[..., 0] + ca * d[..., 1], d[..., 2].expand_as(ca)], -1) + self._static_c0
        v_new = v_new + torch.cat([per_v[..., :2], torch.zeros_like(per_v[..., :1])], -1)
        m = mask.to(dev).bool().reshape(E)
        self.phys.set_static_mesh_points(v_new, m)
        mp0 = mesh_pose[:, 0]
        cb, sb = torch.cos(mp0[:, 3]), torch.sin(mp0[:, 3])
        Rb = torch.zeros(E, 3, 3, device=dev)
        Rb[:, 0, 0], Rb[:, 0, 1], Rb[:, 1, 0], Rb[:, 1, 1], Rb[:, 2, 2] = cb, -sb, sb, cb, 1.0
        shift_b = torch.cat([mp0[:, :2], torch.zeros(E, 1, device=dev)], 1)
        self._box_posed = True
        self._box_c = torch.where(m[:, None], torch.tensor(self._box[0], dtype=torch.float32, device=dev)[None] + shift_b, self._box_c)
        self._box_R = torch.where(m[:, None, None], Rb, self._box_R)

    def reset(self, env_ids=None, episode_ids=None):
        """BaseEnv.reset (env.py:30-51) for some environments of the batch (``env_ids``: indices, a bool mask [n_env], or None =
        all) while the others keep running — episodes are independent and end at different steps (eval_policy_parallel.py:
        266-280).  The reference rebuilds renderer state and a NEW dynamics module per reset (gs_renderer.reset_state,
        phystwin.py:39-102); here the environment's slice of the batch goes back to what the rollout started from: particles at
        rest in their start pose (zero collision forces, grasp state machine at current_openness = None / grasped = False), the
        object's Gaussians and their bones as loaded, the end effector at its start pose; the synthetic action trace of that
        environment starts over.  Everything stays on the device, nothing is synchronised; the candidate lists are rebuilt by
        the next step, and the next ``get_obs()`` renders the reset state before it hands anything out (the reference's reset
        returns get_obs()).

        ``episode_ids`` (a sequence / tensor [n_env]; entries of environments outside ``env_ids`` are ignored; needs
        ``randomize=True``): the reference's ``env.reset(seed=episode_id)`` — the object of a reset environment is placed at the grid
        pose of its episode id (``episode_pose``): particles, Gaussian centres and rotations turned about the object's vertical axis
        and shifted, the resting-pair set of those environments rebuilt from the new positions, the gripper's start pose shifted
        along (and turned with the object for the gripper scenes, whose synthetic trace closes the fingers across the object).
        ``random_variables[episode_id]`` records [x, y, z, angle] like the reference.

        A fault raised by an environment that is NOT being reset survives a partial reset (the next ``step`` reports it); resetting
        every environment hands in a whole new state and clears it."""
        E, dev = self.n_env, self.device
        if env_ids is None:
            mask = torch.ones(E, dtype=torch.bool, device=dev)
            everything = True
        else:
            # a selection that lives on the HOST and names every environment is a full reset (a whole new state: clears a sticky fault
            # word — the episode scheduler always resets through a mask, evaluate.run_episodes); a device-side selection is not read back
            src = env_ids if torch.is_tensor(env_ids) else torch.as_tensor(env_ids)
            everything = False
            if src.device.type == "cpu":
                everything = bool(src.reshape(E).all()) if src.dtype == torch.bool else len(set(src.long().reshape(-1).tolist())) == E
            ids = src.to(dev)
            if ids.dtype == torch.bool:
                mask = ids.reshape(E).clone()
            else:
                mask = torch.zeros(E, dtype=torch.bool, device=dev)
                mask[ids.long().reshape(-1)] = True
        main = torch.cuda.current_stream(dev)
        ev = getattr(self, "_cand_done", None)
        if ev is not None:                      # a candidate rebuild on the side stream is still reading the state about to be replaced
            main.wait_event(ev)
            self._cand_done = None
        self.wait_render()                      # pipelined mode: the render stream reads the Gaussians about to be replaced
        m3 = mask[:, None, None]
        x_new, v_new, means_new = self._init["x"], self._init["v"], self._init["means"]
        rot_new = self._init.get("rot")
        eef_xyz_new, eef_rot_new = self._init.get("eef_xyz"), self._init.get("eef_rot")
        posed = episode_ids is not None
        if posed:
            if not self.randomize:
                raise ValueError("reset(episode_ids=...) needs BatchedRollout(randomize=True)")
            eids = [int(e) for e in (episode_ids.tolist() if torch.is_tensor(episode_ids) else episode_ids)]
            assert len(eids) == E, "episode_ids has one entry per environment"
            if env_ids is None:
                mask_host = [True] * E
            else:       # which rows to fill: from the caller's own (host) ids when they are on the host — no device read then
                src = torch.as_tensor(env_ids).cpu()
                mask_host = src.reshape(E).tolist() if src.dtype == torch.bool else [e in set(src.long().reshape(-1).tolist()) for e in range(E)]
            pose = np.zeros((E, 4), np.float32)
            n_grid_meshes = len(self.GRIDS[self.ob_shape]["meshes"])
            mesh_pose = np.zeros((E, max(n_grid_meshes, 1), 4), np.float32)   # one grid pose per static mesh that has a grid (the k-th grid = the k-th static mesh)
            for e in range(E):
                if mask_host[e]:
                    pose[e] = self.episode_pose(eids[e])
                    self.random_variables[eids[e]] = [float(a) for a in pose[e]]
                    mp = self.episode_mesh_poses(eids[e])
                    if mp:
                        mesh_pose[e, : len(mp)] = mp
                        self.random_mesh_variables[eids[e]] = [[float(a) for a in q] for q in mp]
            pose_t = torch.from_numpy(pose).to(dev)
            if n_grid_meshes and self._static_v0 is not None:
                self._pose_static_meshes(torch.from_numpy(mesh_pose).to(dev), torch.tensor(mask_host, device=dev))
            ca, sa = torch.cos(pose_t[:, 3]), torch.sin(pose_t[:, 3])
            Rz = torch.zeros(E, 3, 3, device=dev)
            Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = ca, -sa, sa, ca, 1.0
            shift = torch.cat([pose_t[:, :2], torch.zeros(E, 1, device=dev)], 1)
            c = self._obj_center
            turn = lambda p: (p - c).matmul(Rz.transpose(1, 2)) + c + shift[:, None]  # noqa: E731  env 0 (unshifted) is the template of every pose
            x_new = turn(self._init["x"][0][None].expand(E, -1, -1))
            v_new = self._init["v"][0][None].expand(E, -1, -1).matmul(Rz.transpose(1, 2))
            means_new = turn(self._init["means"][0][None].expand(E, -1, -1))
            half = 0.5 * pose_t[:, 3]
            qz = torch.stack([torch.cos(half), torch.zeros_like(half), torch.zeros_like(half), torch.sin(half)], 1)      # (w, x, y, z)
            q = self._init["rot"][0][None].expand(E, -1, -1)
            w1, z1 = qz[:, None, 0], qz[:, None, 3]
            rot_new = torch.nn.functional.normalize(torch.stack([w1 * q[..., 0] - z1 * q[..., 3], w1 * q[..., 1] - z1 * q[..., 2],
                                                                 w1 * q[..., 2] + z1 * q[..., 1], w1 * q[..., 3] + z1 * q[..., 0]], -1), dim=-1)
            if self.with_gripper:
                e0 = self._init["eef_xyz"][0][None].expand(E, -1)
                if self.use_pusher:
                    # the rod starts in front of the TURNED block's -x face, at the height of its start pose and the distance the trace
                    # covers until `close_at` (as the constructor places it for the unturned block): the face is wherever the turned
                    # particles reach furthest in -x — a rod start that was only shifted could sit inside a block turned by 45 .. 315 degrees
                    xmin = x_new[:, :, 0].min(1).values
                    near = x_new[:, :, 0] < (xmin[:, None] + 0.01)
                    y_face = torch.where(near, x_new[:, :, 1], torch.full_like(x_new[:, :, 1], float("nan"))).nanmedian(1).values
                    d0 = self.scene["rod_offset_x"]       # host data of the constructor: nothing is read back from the device here
                    eef_xyz_new = torch.stack([xmin + d0, y_face, e0[:, 2]], 1)
                    eef_rot_new = self._init["eef_rot"]
                else:
                    eef_xyz_new = (e0 - c)[:, None].matmul(Rz.transpose(1, 2))[:, 0] + c + shift
                    eef_rot_new = Rz.bmm(self._init["eef_rot"][0][None].expand(E, -1, -1))
        # an UNPOSED reset of environments that an earlier posed reset had re-posed puts object AND obstacle back where the scene was loaded
        # (the reference reloads the default mesh poses with every reset): the box, its boxes in the stepper, the success predicate's
        # box — and the resting-pair set, which belongs to the start positions
        unpose = (not posed) and self._box_posed
        if unpose and self._static_v0 is not None:
            self._pose_static_meshes(torch.zeros(E, 1, 4, device=dev), mask)
        repair = posed or (unpose and getattr(self, "_obj_posed", False))
        if posed:
            self._obj_posed = True
        if everything:
            self.phys.set_state(x_new, v_new)   # a whole new state: clears a sticky fault
            if repair and self.phys.self_collision:
                self.phys.create_resting_case()
        else:
            self.phys.set_state_envs(x_new, v_new, mask, resting_case=repair)
        self.phys.reset_envs(mask)
        obj = self.means[:, : self.n_obj]
        obj.copy_(torch.where(m3, means_new, obj))     # in place: the prepared raster sets point at this storage
        if rot_new is not None:
            ro_ = self.rot_env[:, : self.n_obj]
            ro_.copy_(torch.where(m3, rot_new, ro_))
        self.bones.copy_(torch.where(m3, x_new, self.bones))
        if self.with_gripper:
            self.eef_xyz = torch.where(mask[:, None], eef_xyz_new, self.eef_xyz)
            self.eef_rot = torch.where(m3, eef_rot_new, self.eef_rot)
            self.eef_gripper = torch.where(mask, torch.ones_like(self.eef_gripper), self.eef_gripper)
        self._env_t0 = torch.where(mask, torch.full_like(self._env_t0, self.t), self._env_t0)
        self._restarted = True
        self._cand_fresh = False
        self._frame_dirty = True

    # ---- one batched env step -----------------------------------------------------------------------------------
    def physics_step(self, action=None):
        # update_collision_graph works on the positions the previous step left (phystwin.py:365-366 calls it first thing in
        # step()); it is enqueued right AFTER the previous physics graph instead, so that its candidate count — which picks the
        # graph flavour on the host — has long landed when the next step starts, and the host never waits for the GPU
        if self.phys.self_collision and not self._cand_fresh:
            self.phys.update_collision_graph()
        ev = getattr(self, "_cand_done", None)
        if ev is not None:                                  # the rebuild enqueued next to the previous step's rendering (below)
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._cand_done = None
        if self.with_gripper:
            self.apply_action(action if action is not None else self.synthetic_action(self.t))
        lg = self._log
        if lg is not None and lg["i"] < lg["n"]:
            lg["p0"][lg["i"]].record()
        self.phys.step(0, 0, sync_state=True)
        if lg is not None and lg["i"] < lg["n"]:
            lg["p1"][lg["i"]].record()
            self.phys.log_contacts(lg["counts"][lg["i"]])
            lg["flavour"].append(self.phys.last_flavour()["kernel"] + f" x{self.phys.last_flavour()['chains']} chains")
        if self.phys.self_collision:
            # ... on a second stream: the rebuild (spatial sort + candidate lists: 0.1 ms for one environment, 0.3 ms for 32) only reads the
            # state this step left, like the skinning + rasterisation that follow on the launch stream, and nothing before the next
            # step's substeps needs its result — so the two run side by side instead of one after the other
            if os.environ.get("R2S_CAND_STREAM", "1") == "0":     # knob: rebuild on the launch stream, before the rendering
                self.phys.update_collision_graph()
                self._cand_fresh = True
                return
            main = torch.cuda.current_stream(self.device)
            if getattr(self, "_cand_stream", None) is None:
                self._cand_stream = self.phys.side_stream(1)   # a pooled stream of the library, idle between env steps: a stream of our own
                                                               # would be one more hardware queue, and the next step's chains would share one
            stepped = torch.cuda.Event()
            stepped.record(main)
            self._cand_stream.wait_event(stepped)
            with torch.cuda.stream(self._cand_stream):
                self.phys.update_collision_graph()
                self._cand_done = torch.cuda.Event()
                self._cand_done.record(self._cand_stream)
            self._cand_fresh = True

    def check_fault(self):
        """Raise now if a kernel of an earlier step declared the state invalid (PhysBatch.check_fault: waits for the stream) — what the
        episode scheduler asks before it records episodes whose end is followed by a full reset (evaluate.run_episodes)."""
        self.phys.check_fault()

    # ---- per-step log of a timed window: stamps on the launch stream + contact counters kept on the device ------------
    def start_log(self, n_steps):
        ev = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]  # noqa: E731
        self._log = dict(n=n_steps, i=0, s=ev(), p0=ev(), p1=ev(), flavour=[],
                         counts=torch.zeros(n_steps, 3, dtype=torch.int32, device=self.device))
        self._log["s"][0].record()

    def read_log(self):
        lg, self._log = self._log, None
        torch.cuda.synchronize(self.device)
        n = lg["i"]
        c = lg["counts"].cpu().numpy()
        return dict(step_ms=[lg["s"][k].elapsed_time(lg["s"][k + 1]) for k in range(n)],
                    phys_ms=[lg["p0"][k].elapsed_time(lg["p1"][k]) for k in range(n)],
                    candidates=c[:n, 0].tolist(), mesh_hits=c[:n, 1].tolist(), grasped=c[:n, 2].tolist(), flavour=lg["flavour"][:n])

    def contact_stats(self):
        """What the last physics step touched: particles with self-collision candidates, particles that reacted to a
        collision mesh in the last substep, environments whose grasp state machine holds the object, and the captured
        kernel flavour that ran."""
        n_cand, hits = self.phys.contact_stats()
        grasped = 0
        if self.with_gripper and not self.use_pusher:
            grasped = int(self.phys.eef_state()[1].sum().item())
        return dict(self_collision_candidates=int(n_cand), mesh_contacts=int(hits), grasped_envs=grasped, flavour=self.phys.last_flavour())

    def _update_cameras(self):
        """Wrist camera of every environment from its current end-effector pose (gs_renderer.py:966-985), on the device."""
        if self.wrist is not None:
            if self.with_gripper:
                self.wrist.update(self.eef_xyz, self.eef_rot)
            else:
                self.wrist.update(self._wrist_xyz, self._wrist_rot)

    def _poll_raster(self, wait=False):
        running, n, ov = self.raster.poll(wait)
        if ov > self._raster_overflows:
            self.lossy_batches += ov - self._raster_overflows
            self._raster_overflows = ov
            return True
        if not running:
            self.last_num_rendered = n
        return False

    def render(self):
        self._update_means()
        self._update_cameras()
        if self._prepared is None:
            self._prepared = self.raster.prepare(self._sets, self._frames)
        self.last_num_rendered = self.raster.forward(self._prepared, None, self.W, self.H)
        self._poll_raster()          # non-blocking: an EARLIER batch that overflowed its capacity is counted in lossy_batches
        self._frame_dirty = False
        return self.out_color, self.out_depth

    def observations(self):
        """The images of the last step as a closed-loop caller may read them: waits for the render, and if that batch overflowed
        the capacity of the sync-free pipeline (its deepest instances were dropped) renders the step again — the next forward
        re-sizes by reading the count once — so what is returned is always the complete frame.  (`lossy_batches` counts how often
        that happened; an open-loop caller that only reads ``out_color`` after a device synchronisation should look at it.)"""
        self.wait_render()
        if self._frame_dirty:                   # a reset since the last render (or nothing rendered yet): the frame of the state as it is now
            self.render()
        if self._poll_raster(wait=True):
            self.last_num_rendered = self.raster.forward(self._prepared, None, self.W, self.H)
            self._poll_raster(wait=True)
        return self.out_color, self.out_depth

    def robot_state(self):
        """obs['robot'] of BaseEnv.get_obs (env.py:62-66) for every environment, device tensors: eef_xyz [E,3], eef_quat [E,4]
        (w, x, y, z of the current end-effector rotation, phystwin.py:117), eef_gripper [E,1] (the opening commanded by the last
        action; 1 = open)."""
        if self.eef_rot.is_cuda:      # one launch (r2s_rot_to_quat) instead of the ~40 of the torch restatement: 0.2 ms of host time per observation
            from .camera import rot_to_quat
            quat = rot_to_quat(self.eef_rot)
        else:
            quat = rotation_matrix_to_quaternion(self.eef_rot)
        return dict(eef_xyz=self.eef_xyz, eef_quat=quat, eef_gripper=self.eef_gripper[:, None])

    def get_obs(self):
        """BaseEnv.get_obs (env.py:53-74) for the batch: the fixed-camera and wrist-camera images of the last step — complete and
        validated (``observations``), views of the output arrays, [E,3,H,W] / [E,1,H,W] per camera — and the robot state."""
        col, dep = self.observations()
        wrist = [1] if self.wrist is not None else []
        fixed = [v for v in range(self.views) if v not in wrist]
        return dict(image_list=[col[:, v] for v in fixed], depth_list=[dep[:, v] for v in fixed],
                    image_wrist_list=[col[:, v] for v in wrist], depth_wrist_list=[dep[:, v] for v in wrist],
                    image_extra=None, depth_extra=None, robot=self.robot_state() if self.with_gripper else None)

    def camera_numpy(self, e, v):
        """Settings of view ``v`` of environment ``e`` as the rasteriser currently sees them (numpy; parity tests, cpu_baseline)."""
        cam = dict(self.cams[v])
        if v == 1 and self.wrist is not None:
            torch.cuda.synchronize(self.device)
            cam["viewmatrix"] = self.wrist.viewmatrix[e].cpu().numpy()
            cam["projmatrix"] = self.wrist.projmatrix[e].cpu().numpy()
            cam["campos"] = self.wrist.campos[e].cpu().numpy()
        return cam

    # ---- throughput mode for open-loop stretches (an action chunk, a replay): render(t) overlaps physics(t+1) ------------
    def set_pipelined(self, on: bool):
        """With ``on``, ``step`` enqueues skinning + rasterisation of env step t on a second stream and returns at once; the
        physics of step t+1 only waits for the (short) skinning kernels, which are the readers of the particle state, so the
        raster pipeline runs next to the next step's substeps.  The images of step t are complete after ``wait_render()``
        (or any device synchronisation).  Same kernels on the same data: results are identical to the serial mode (tested).
        Meaningful when the next action does not depend on this step's observation — inside an action chunk of the policy —
        and therefore NOT what ``bench.py``'s ``value`` measures (that is the closed loop: observation before next action)."""
        self._pipelined = bool(on)
        if on and getattr(self, "_render_stream", None) is None:
            self._render_stream = torch.cuda.Stream(device=self.device)
        if not on:
            self.wait_render()

    def wait_render(self):
        ev = getattr(self, "_render_done", None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _render_pipelined(self):
        main = torch.cuda.current_stream(self.device)
        rs = self._render_stream
        ready = torch.cuda.Event()
        ready.record(main)                       # physics(t) has written x
        rs.wait_event(ready)
        with torch.cuda.stream(rs):
            self._update_means()
            self._update_cameras()
            skinned = torch.cuda.Event()
            skinned.record(rs)
            if self._prepared is None:
                self._prepared = self.raster.prepare(self._sets, self._frames)
            self.last_num_rendered = self.raster.forward(self._prepared, None, self.W, self.H)
            self._poll_raster()
            self._frame_dirty = False
            self._render_done = torch.cuda.Event()
            self._render_done.record(rs)
        main.wait_event(skinned)                 # the next step's state write-back must not overtake the skinning that reads x
        return self.out_color, self.out_depth

    def step(self, action=None):
        """One batched env step, ENQUEUED: ``action``: None (the synthetic trace), a dict of per-environment motion tensors, or an
        [n_env, 13] 'xyz_rot' action tensor — see ``apply_action``.  Nothing is waited for.  Returns ``(out_color, out_depth)`` — the
        output arrays the step's frames are being rendered into, as every round before round 4 did — UNVALIDATED: the kernels that
        fill them are only enqueued, and a sync-free raster batch that overflowed its capacity leaves its deepest instances out
        (``lossy_batches``).  What a closed loop reads is ``get_obs()`` / ``observations()``: they wait for the render, check the
        batch and re-render a lossy one.  ``enqueue_step`` is the same method under the name that says so."""
        self.physics_step(action)
        if getattr(self, "_pipelined", False):
            out = self._render_pipelined()
        else:
            out = self.render()
        self.t += 1
        lg = self._log
        if lg is not None and lg["i"] < lg["n"]:
            lg["i"] += 1
            lg["s"][lg["i"]].record()
        return out

    enqueue_step = step

    # ---- the Gaussian cloud of one environment as the rasteriser currently sees it (parity tests) ----------------------
    def g_env(self, e):
        if self.rot_env is None:
            return self.g
        return dict(self.g, rotations=self.rot_env[e])

    def scene_numpy(self, e):
        sc = {k: v.cpu().numpy() for k, v in self.g_env(e).items()}
        sc["means3D"] = self.means[e].cpu().numpy()
        return sc

    # ---- task success of the current state, on the device (calculate_success_{rope,sloth,T}.py) -----------------------
    def success_flags(self):
        """bool [n_env]: the frame-level predicate of the scene's task, evaluated on the device-resident particle state."""
        from . import metrics

        x = self.phys.x
        if self.ob_shape == "rope":
            return metrics.rope_routed(x, self._springs_dev)
        if self.ob_shape == "T":
            return metrics.pusht_success(x, self._target_dev)
        # sloth: >= 3050 of ~15k particles (the same fraction here) inside the box obstacle's OBB scaled by 1.05
        c, half = self._box
        need = int(round(3050 / 15000 * self.N))
        if self._box_posed:     # boxes re-posed per environment by posed resets: the same count against each environment's own box
            loc = (x - self._box_c[:, None]).matmul(self._box_R)          # box axes = columns of R
            inside = (loc.abs() <= torch.tensor(1.05 * half, dtype=torch.float32, device=x.device)).all(-1)
            return inside.sum(1) >= need
        return metrics.points_in_obb(x, c, np.eye(3), 1.05 * half) >= need

    # ---- accounting -------------------------------------------------------------------------------
    def physics_algorithmic_bytes_per_substep(self):
        return self.n_env * (16 * self.S + 48 * self.N)

    def composite_algorithmic_bytes(self, num_rendered=None):
        L = self.last_num_rendered if num_rendered is None else num_rendered
        return 44 * L + 16 * self.W * self.H * self.n_env * self.views
