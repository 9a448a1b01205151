"""Generates tests/golden/robot_gs_{gripper,pusher}.npz by RUNNING THE REFERENCE's own robot-Gaussian placement
    transform_gs_xarm_gripper / transform_gs_xarm_pusher   (/root/reference/sim/utils/robot/robot_pc_transformations.py:12-55, :94-140)
    RobotPcSampler.transform_gs_torch + quat_mult_torch      (/root/reference/sim/utils/robot/robot_pc_sampler.py:17-24, :118-161)
on the CPU in the authoring container.  What is NOT the reference's code in this run, and why:
  * SAPIEN forward kinematics (absent here, and out of scope: the device kernel takes per-link poses as input): the sampler
    object is created without __init__ and given a stand-in `robot_model` whose get_link_pose(i) returns the rigid matrices
    recorded in the fixture as `link_pose` / `link_pose_base`;
  * kornia.geometry.conversions.rotation_matrix_to_quaternion (third party, absent): restated below from kornia's published
    source (0.7.x, (w, x, y, z) order, eps = 1e-8) — this one conversion per link is therefore NOT pinned by the fixture;
  * open3d / urdfpy / transforms3d / sapien imports of the two modules: empty placeholder modules (unused by these functions).
Everything else — matrix composition with the URDF offsets, the inverse of the base pose, the point transform, quat_mult_torch,
the per-link masking and scatter, F.normalize of the scan's rotations — is executed from the reference's files.  Data only.

Usage (authoring container only):  python tests/golden/make_robot_gs_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def kornia_rotation_matrix_to_quaternion(rotation_matrix, eps=1e-8):
    m = rotation_matrix.reshape(*rotation_matrix.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(m, chunks=9, dim=-1)
    trace = m00 + m11 + m22

    def safe_zero_division(a, b):
        return a / torch.clamp(b, min=torch.finfo(b.dtype).tiny)

    def trace_positive_cond():
        sq = torch.sqrt(trace + 1.0 + eps) * 2.0
        return torch.cat((0.25 * sq, safe_zero_division(m21 - m12, sq), safe_zero_division(m02 - m20, sq), safe_zero_division(m10 - m01, sq)), dim=-1)

    def cond_1():
        sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
        return torch.cat((safe_zero_division(m21 - m12, sq), 0.25 * sq, safe_zero_division(m01 + m10, sq), safe_zero_division(m02 + m20, sq)), dim=-1)

    def cond_2():
        sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
        return torch.cat((safe_zero_division(m02 - m20, sq), safe_zero_division(m01 + m10, sq), 0.25 * sq, safe_zero_division(m12 + m21, sq)), dim=-1)

    def cond_3():
        sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
        return torch.cat((safe_zero_division(m10 - m01, sq), safe_zero_division(m02 + m20, sq), safe_zero_division(m12 + m21, sq), 0.25 * sq), dim=-1)

    where_2 = torch.where(m11 > m22, cond_2(), cond_3())
    where_1 = torch.where((m00 > m11) & (m00 > m22), cond_1(), where_2)
    return torch.where(trace > 0.0, trace_positive_cond(), where_1)


def load_reference():
    for name in ("open3d", "urdfpy", "sapien", "sapien.core", "transforms3d", "kornia", "kornia.geometry", "kornia.geometry.conversions"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["urdfpy"].URDF = object
    sys.modules["sapien"].core = sys.modules["sapien.core"]
    k = sys.modules["kornia"]
    k.geometry = sys.modules["kornia.geometry"]
    k.geometry.conversions = sys.modules["kornia.geometry.conversions"]
    k.geometry.conversions.rotation_matrix_to_quaternion = kornia_rotation_matrix_to_quaternion
    sys.path.insert(0, "/root/reference")
    import sim.utils.robot.robot_pc_sampler as sampler
    import sim.utils.robot.robot_pc_transformations as tf
    return sampler, tf


def rigid(rng, scale=1.0):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    th = rng.uniform(-np.pi, np.pi) * scale
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    M = np.eye(4)
    M[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    M[:3, 3] = rng.uniform(-0.4, 0.4, 3) * scale + np.array([0.3, 0.0, 0.3])
    return M


class FakePose:
    def __init__(self, m): self.m = m
    def to_transformation_matrix(self): return self.m


class FakeModel:
    """Stands where SAPIEN's pinocchio model stands: poses are looked up by the qpos handed to compute_forward_kinematics."""
    def __init__(self, table): self.table, self.cur = table, None
    def compute_forward_kinematics(self, qpos): self.cur = self.table[tuple(np.round(np.asarray(qpos, np.float64), 9))]
    def get_link_pose(self, idx): return FakePose(self.cur[idx])


class FakeLink:
    def __init__(self, name): self.name = name


def run(kind, seed, n_scan, n_cfg=3):
    sampler, tf = load_reference()
    rng = np.random.default_rng(seed)
    n_links = 18 if kind == "gripper" else 11
    link_ids = [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16] if kind == "gripper" else None
    links = [FakeLink(f"link_{i}") for i in range(n_links)]
    rob = object.__new__(sampler.RobotPcSampler)
    rob.sapien_robot = types.SimpleNamespace(get_links=lambda: links)
    rob.offsets = {l.name: rigid(rng, 0.1) - np.diag([0, 0, 0, 0]) for l in links}
    init_qpos = [0, -45, 0, 30, 0, 75, 0]
    out = dict(kind=kind)
    # the scan: robot + table Gaussians with a per-Gaussian link id (total_mask), unnormalised rotations
    means = rng.uniform(-0.5, 0.5, (n_scan, 3)).astype(np.float32)
    quats = rng.normal(size=(n_scan, 4)).astype(np.float32) * rng.uniform(0.5, 2.0, (n_scan, 1)).astype(np.float32)
    total_mask = rng.integers(-1, n_links, n_scan).astype(np.int64)        # -1 and the unlisted link ids stay static
    table = {}

    def fk_for(q):
        key = tuple(np.round(np.asarray(q, np.float64), 9))
        if key not in table:
            table[key] = [rigid(np.random.default_rng(abs(hash(key)) % (2**32) + i)).astype(np.float32) for i in range(n_links)]
        return key

    rob.robot_model = FakeModel(table)
    poses, base_pose, new_means, new_quats, qposes, grips = [], None, [], [], [], []
    for c in range(n_cfg):
        qpos = rng.uniform(-1.0, 1.0, 7)
        grip = float(rng.uniform(0, 800))
        if kind == "gripper":
            full = np.array(list(qpos) + [(800 - grip) * 0.001] * 6)
            base = np.array(list(np.array(init_qpos) * np.pi / 180) + [(800 - 750) * 0.001] * 6)
        else:
            full = np.array(list(qpos))
            base = np.array(init_qpos) * np.pi / 180
        kq, kb = fk_for(full), fk_for(base)
        params = dict(means3D=torch.from_numpy(means.copy()), rotations=torch.from_numpy(quats.copy()))
        if kind == "gripper":
            res = tf.transform_gs_xarm_gripper(qpos, grip, params, torch.from_numpy(total_mask), sample_robot=rob)
        else:
            res = tf.transform_gs_xarm_pusher(qpos, params, torch.from_numpy(total_mask), sample_robot=rob)
        poses.append(np.stack(table[kq])); base_pose = np.stack(table[kb])
        new_means.append(res["means3D"].numpy().copy()); new_quats.append(res["rotations"].numpy().copy())
        qposes.append(qpos); grips.append(grip)
    out.update(means=means, quats=quats, total_mask=total_mask.astype(np.int32), offsets=np.stack([rob.offsets[l.name] for l in links]).astype(np.float64),
               link_pose=np.stack(poses).astype(np.float32), link_pose_base=base_pose.astype(np.float32), new_means=np.stack(new_means),
               new_quats=np.stack(new_quats), n_links=n_links)
    return out


def main():
    g = run("gripper", 0, 4000)
    moved = np.abs(g["new_means"][0] - g["means"]).max(1) > 0
    print("gripper: links", g["n_links"], "moved", int(moved.sum()), "of", len(moved), "static", int((~moved).sum()))
    np.savez_compressed(os.path.join(HERE, "robot_gs_gripper.npz"), **{k: v for k, v in g.items() if k != "kind"})
    try:
        p = run("pusher", 1, 3000)
        moved = np.abs(p["new_means"][0] - p["means"]).max(1) > 0
        print("pusher: links", p["n_links"], "moved", int(moved.sum()), "of", len(moved))
        np.savez_compressed(os.path.join(HERE, "robot_gs_pusher.npz"), **{k: v for k, v in p.items() if k != "kind"})
    except Exception as e:  # the pusher variant asserts its own link count
        print("pusher variant not generated:", repr(e))


if __name__ == "__main__":
    main()
