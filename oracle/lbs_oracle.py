"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

numpy restatement of the reference's linear-blend skinning step `interpolate_motions`
(sim/utils/gs/transform_utils.py:58-212) for the way the simulator calls it (quat=None, precomputed weights;
sim/renderer/gs_renderer.py:738-747).  PINNED: tests/test_lbs_oracle.py checks it against fixtures produced by running
the reference function itself (tests/golden/make_lbs_golden.py -> tests/golden/lbs_*.npz).
"""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)


def bone_rotations(bones, motions, relations):
    """Per-bone rotation of transform_utils.py:73-165: Kabsch fit of the bone's k neighbours before / after the motion.

    F = sum_k (a'_k)(a_k)^T (:83); rank by torch.linalg.matrix_rank's default tolerance max(m,n)*eps*sigma_max (:85);
    rank 2 or 3: R = U S V^T with S flipped so that det R = +1 (:96-114), i.e. the proper rotation closest to F.
    If ANY bone has rank < 2 the reference's assignment `bone_transforms[:, :3, :3] = R` (:157) fails on the shape
    mismatch and its `except` branch sets every rotation to the identity (:159-162); reproduced here."""
    bones = np.asarray(bones, np.float32); motions = np.asarray(motions, np.float32)
    rel = np.asarray(relations)
    adj = (bones[rel] - bones[:, None]).astype(np.float32)                                       # :79
    adj_new = ((bones[rel] + motions[rel]) - (bones[:, None] + motions[:, None])).astype(np.float32)  # :80
    F = np.einsum("nki,nkj->nij", adj_new, adj).astype(np.float32)                                # :83  (n,3,3)
    U, S, Vt = np.linalg.svd(F.astype(np.float64))
    rank = (S > (3 * EPS32 * S[:, :1])).sum(1)
    if not np.all(rank >= 2):
        return np.tile(np.eye(3, dtype=np.float32), (len(bones), 1, 1)), rank
    d = np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    D = np.tile(np.eye(3), (len(bones), 1, 1)); D[:, 2, 2] = d
    return (U @ D @ Vt).astype(np.float32), rank


def interpolate_motions(bones, motions, relations, xyz, weights, weights_indices):
    """xyz' = sum_j w_j (R_bj (x - b_j) + m_j + b_j), transform_utils.py:178-189."""
    R, _ = bone_rotations(bones, motions, relations)
    bones = np.asarray(bones, np.float32); motions = np.asarray(motions, np.float32); xyz = np.asarray(xyz, np.float32)
    wi = np.asarray(weights_indices); w = np.asarray(weights, np.float32)
    bp = bones[wi]                                   # (P,k,3)
    t = xyz[:, None] - bp
    t = np.einsum("pkij,pkj->pki", R[wi], t).astype(np.float32)
    t = t + motions[wi] + bp
    return (t * w[:, :, None]).sum(1).astype(np.float32)


def knn_relations(bones, k=8):
    """knn_relations, sim/renderer/gs_renderer.py:195-200 (kd-tree k nearest bones, self excluded)."""
    from scipy.spatial import cKDTree

    b = np.asarray(bones, np.float64)
    _, idx = cKDTree(b).query(b, k=k + 1)
    return idx[:, 1:].astype(np.int32)


def knn_weights(bones, pts, k=16):
    """knn_weights, sim/renderer/gs_renderer.py:202-211 (inverse-distance weights over the k nearest bones)."""
    from scipy.spatial import cKDTree

    b = np.asarray(bones, np.float32); p = np.asarray(pts, np.float32)
    _, idx = cKDTree(b.astype(np.float64)).query(p.astype(np.float64), k=k)
    dist = np.linalg.norm(b[idx] - p[:, None], axis=-1).astype(np.float32)
    w = (1.0 / (dist + np.float32(1e-6))).astype(np.float32)
    w = w / w.sum(-1, keepdims=True)
    return w.astype(np.float32), idx.astype(np.int32)


def rotate_quats(bones, motions, relations, quat, weights, weights_indices):
    """The quat branch of interpolate_motions (transform_utils.py:197-210): bone rotations -> unit quaternions (kornia's
    rotation_matrix_to_quaternion, restated in oracle/robot_oracle.py; third party: that conversion is not pinned), blended with
    the skinning weights, normalised, composed with the splat's quaternion (blended rotation first)."""
    from .robot_oracle import rotation_matrix_to_quaternion

    R, _ = bone_rotations(bones, motions, relations)
    bq = np.stack([np.asarray(rotation_matrix_to_quaternion(r), np.float32) for r in R.astype(np.float32)])
    bq = bq / np.maximum(np.linalg.norm(bq, axis=-1, keepdims=True), 1e-12)
    w = np.asarray(weights, np.float32)
    q = (bq[np.asarray(weights_indices)] * w[:, :, None]).sum(1).astype(np.float32)
    q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-12)
    p = np.asarray(quat, np.float32)
    return np.stack([q[:, 0] * p[:, 0] - q[:, 1] * p[:, 1] - q[:, 2] * p[:, 2] - q[:, 3] * p[:, 3],
                     q[:, 0] * p[:, 1] + q[:, 1] * p[:, 0] + q[:, 2] * p[:, 3] - q[:, 3] * p[:, 2],
                     q[:, 0] * p[:, 2] - q[:, 1] * p[:, 3] + q[:, 2] * p[:, 0] + q[:, 3] * p[:, 1],
                     q[:, 0] * p[:, 3] + q[:, 1] * p[:, 2] - q[:, 2] * p[:, 1] + q[:, 3] * p[:, 0]], -1).astype(np.float32)
