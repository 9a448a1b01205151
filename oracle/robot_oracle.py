"""CPU restatement (numpy, float32) of the reference's robot-Gaussian placement — TEST INFRASTRUCTURE, never imported by the
product package.

    transform_gs_torch + quat_mult_torch   sim/utils/robot/robot_pc_sampler.py:17-24, :118-161
    transform_gs_xarm_gripper / _pusher    sim/utils/robot/robot_pc_transformations.py:12-55, :94-133
    final F.normalize of every rotation     sim/renderer/gs_renderer.py:906

Pinned by tests/golden/robot_gs_{gripper,pusher}.npz, which the reference's own functions produced
(tests/golden/make_robot_gs_golden.py) — except kornia's rotation_matrix_to_quaternion, which is third-party code absent from
/root/reference and this image: restated here from its published source (kornia 0.7: (w, x, y, z), eps = 1e-8) and used by the
fixture generator in the same form, so that one conversion is "parity unpinned"."""
import numpy as np

GRIPPER_LINKS = [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16]   # robot_pc_transformations.py:33 (of 18 links)
PUSHER_LINKS = [1, 2, 3, 4, 5, 6, 7, 8, 10]                             # :113 (of 11 links)
f32 = np.float32


def rotation_matrix_to_quaternion(R, eps=1e-8):
    """kornia.geometry.conversions.rotation_matrix_to_quaternion for one 3x3 float32 matrix -> (w, x, y, z)."""
    R = np.asarray(R, f32)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [f32(v) for v in R.reshape(9)]
    tiny = np.finfo(f32).tiny
    div = lambda a, b: f32(a) / max(f32(b), tiny)  # noqa: E731
    trace = f32(f32(m00 + m11) + m22)
    if trace > 0:
        sq = f32(np.sqrt(f32(f32(trace + f32(1.0)) + f32(eps))) * f32(2.0))
        q = [f32(0.25) * sq, div(m21 - m12, sq), div(m02 - m20, sq), div(m10 - m01, sq)]
    elif m00 > m11 and m00 > m22:
        sq = f32(np.sqrt(f32(f32(f32(f32(1.0) + m00) - m11) - m22 + f32(eps))) * f32(2.0))
        q = [div(m21 - m12, sq), f32(0.25) * sq, div(m01 + m10, sq), div(m02 + m20, sq)]
    elif m11 > m22:
        sq = f32(np.sqrt(f32(f32(f32(f32(1.0) + m11) - m00) - m22 + f32(eps))) * f32(2.0))
        q = [div(m02 - m20, sq), div(m01 + m10, sq), f32(0.25) * sq, div(m12 + m21, sq)]
    else:
        sq = f32(np.sqrt(f32(f32(f32(f32(1.0) + m22) - m00) - m11 + f32(eps))) * f32(2.0))
        q = [div(m10 - m01, sq), div(m02 + m20, sq), div(m12 + m21, sq), f32(0.25) * sq]
    return np.asarray(q, f32)


def quat_mult(q1, q2):
    """quat_mult_torch, robot_pc_sampler.py:17-24; q1 [4] or [n,4], q2 [n,4]."""
    q1 = np.asarray(q1, f32).reshape(-1, 4); q2 = np.asarray(q2, f32).reshape(-1, 4)
    w1, x1, y1, z1 = q1[:, 0], q1[:, 1], q1[:, 2], q1[:, 3]
    w2, x2, y2, z2 = q2[:, 0], q2[:, 1], q2[:, 2], q2[:, 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1).astype(f32)


def f_normalize(q):
    """torch.nn.functional.normalize(q, dim=-1)."""
    q = np.asarray(q, f32)
    return (q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True).astype(f32), f32(1e-12))).astype(f32)


def link_matrices(link_pose, link_pose_base, offsets):
    """mat_l = (pose_l @ off_l) @ inv(base_l @ off_l), float32 (robot_pc_sampler.py:143-149) and its quaternion (:151)."""
    L = len(offsets)
    off = np.asarray(offsets, np.float64).astype(f32)
    mats, quats = np.zeros((L, 4, 4), f32), np.zeros((L, 4), f32)
    for l in range(L):
        mat = np.asarray(link_pose[l], f32) @ off[l]
        mat_base = np.asarray(link_pose_base[l], f32) @ off[l]
        mats[l] = (mat @ np.linalg.inv(mat_base).astype(f32)).astype(f32)
        quats[l] = rotation_matrix_to_quaternion(mats[l][:3, :3])
    return mats, quats


def transform_gs(means, rotations, total_mask, link_ids, link_pose, link_pose_base, offsets, final_normalize=False):
    """The scan after transform_gs_xarm_* for ONE environment: (means' [n,3], rotations' [n,4])."""
    means = np.asarray(means, f32).copy()
    quats = f_normalize(rotations)                                  # robot_pc_transformations.py:29 — the whole scan
    mats, lq = link_matrices(link_pose, link_pose_base, offsets)
    mask = np.asarray(total_mask)
    for i in link_ids:
        sel = mask == i
        means[sel] = (means[sel] @ mats[i][:3, :3].T + mats[i][:3, 3]).astype(f32)   # :150
        quats[sel] = quat_mult(lq[i], quats[sel])                                      # :153
    return means, (f_normalize(quats) if final_normalize else quats)
