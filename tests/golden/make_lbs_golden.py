"""Generates tests/golden/lbs_*.npz by RUNNING THE REFERENCE's own `interpolate_motions`
(/root/reference/sim/utils/gs/transform_utils.py:58-212) on seeded inputs, on the CPU, in the authoring container.

The reference module imports `kornia` and `diff_gaussian_rasterization` at the top (transform_utils.py:3-4); neither is
installed here and neither is touched by `interpolate_motions` when `quat=None` (the only way the simulator calls it,
sim/renderer/gs_renderer.py:738-747).  The two imports are satisfied with EMPTY placeholder modules so that the module
body can be executed; every arithmetic operation recorded in the fixtures is the reference's own torch code.
Fixtures are data only (inputs + outputs); nothing of the reference's source is stored.

Usage (authoring container only — /root/reference does not exist on the GPU box):  python tests/golden/make_lbs_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/sim/utils/gs/transform_utils.py"


def load_reference():
    for name in ("kornia", "kornia.geometry", "kornia.geometry.conversions", "diff_gaussian_rasterization"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "diff_gaussian_rasterization":
                m.GaussianRasterizationSettings = object
            sys.modules[name] = m
    # quat=... (the `lbs_quat` fixture only) calls kornia.geometry.conversions.rotation_matrix_to_quaternion (third party, absent):
    # restated from kornia's published source in make_robot_gs_golden.py — that one conversion is NOT pinned by the fixture
    from make_robot_gs_golden import kornia_rotation_matrix_to_quaternion
    k = sys.modules["kornia"]
    k.geometry = sys.modules["kornia.geometry"]
    k.geometry.conversions = sys.modules["kornia.geometry.conversions"]
    k.geometry.conversions.rotation_matrix_to_quaternion = kornia_rotation_matrix_to_quaternion
    spec = importlib.util.spec_from_file_location("ref_transform_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def knn(points, queries, k, drop_self):
    from scipy.spatial import cKDTree

    _, idx = cKDTree(points).query(queries, k=k + (1 if drop_self else 0))
    return idx[:, 1:] if drop_self else idx


def case(seed, n_bones, n_pts, k_rel=8, k_wgt=16, mode="smooth"):
    rng = np.random.default_rng(seed)
    bones = rng.uniform(-0.05, 0.05, (n_bones, 3)).astype(np.float32)
    if mode == "smooth":      # rotation about an axis + bending + translation: what a soft body does in one env step
        w = rng.normal(0, 0.4, 3)
        mot = np.cross(w, bones) + 0.3 * bones[:, [1, 2, 0]] ** 2 + rng.normal(0, 0.01, 3)
    elif mode == "noisy":     # adds per-bone noise: exercises ill-conditioned / reflected fits (det F < 0)
        w = rng.normal(0, 1.0, 3)
        mot = np.cross(w, bones) + rng.normal(0, 0.004, bones.shape)
    elif mode == "planar":    # all bones in a plane -> rank-2 covariances everywhere
        bones[:, 2] = 0.0
        w = np.array([0.0, 0.0, 0.7])
        mot = np.cross(w, bones) + np.array([0.01, 0.0, 0.02])
    mot = mot.astype(np.float32)
    xyz = (bones[rng.integers(0, n_bones, n_pts)] + rng.normal(0, 0.003, (n_pts, 3))).astype(np.float32)
    relations = knn(bones.astype(np.float64), bones.astype(np.float64), k_rel, True).astype(np.int64)
    # knn_weights, sim/renderer/gs_renderer.py:202-211 (float32 torch ops)
    tb, tx = torch.from_numpy(bones), torch.from_numpy(xyz)
    dist = torch.norm(tx[:, None] - tb, dim=-1)
    _, indices = torch.topk(dist, k_wgt, dim=-1, largest=False)
    dist = torch.norm(tb[indices] - tx[:, None], dim=-1)
    weights = 1 / (dist + 1e-6)
    weights = weights / weights.sum(dim=-1, keepdim=True)
    return bones, mot, relations, xyz, weights.numpy(), indices.numpy()


def main():
    ref = load_reference()
    for name, args in dict(smooth=(0, 300, 1200, 8, 16, "smooth"), noisy=(1, 256, 900, 8, 16, "noisy"),
                           planar=(2, 200, 500, 8, 16, "planar"), small_k=(3, 120, 400, 4, 5, "smooth")).items():
        bones, mot, rel, xyz, w, wi = case(*args)
        out, _, _ = ref.interpolate_motions(bones=torch.from_numpy(bones), motions=torch.from_numpy(mot), relations=torch.from_numpy(rel),
                                            xyz=torch.from_numpy(xyz), quat=None, weights=torch.from_numpy(w),
                                            weights_indices=torch.from_numpy(wi), device="cpu")
        np.savez_compressed(os.path.join(HERE, f"lbs_{name}.npz"), bones=bones, motions=mot, relations=rel.astype(np.int32), xyz=xyz,
                            weights=w.astype(np.float32), weights_indices=wi.astype(np.int32), xyz_out=out.numpy().astype(np.float32))
        print(name, "max |dx|", float(np.abs(out.numpy() - xyz).max()))
    # the quat path (transform_utils.py:197-210): splats rotated by the blended bone rotations
    bones, mot, rel, xyz, w, wi = case(5, 260, 1000, 8, 16, "smooth")
    rng = np.random.default_rng(55)
    quat = rng.normal(size=(len(xyz), 4)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    out, rot, _ = ref.interpolate_motions(bones=torch.from_numpy(bones), motions=torch.from_numpy(mot), relations=torch.from_numpy(rel),
                                          xyz=torch.from_numpy(xyz), quat=torch.from_numpy(quat), weights=torch.from_numpy(w),
                                          weights_indices=torch.from_numpy(wi), device="cpu")
    np.savez_compressed(os.path.join(HERE, "lbs_quat.npz"), bones=bones, motions=mot, relations=rel.astype(np.int32), xyz=xyz, quat=quat,
                        weights=w.astype(np.float32), weights_indices=wi.astype(np.int32), xyz_out=out.numpy().astype(np.float32),
                        quat_out=rot.numpy().astype(np.float32))
    print("quat: max |dq|", float(np.abs(rot.numpy() - quat).max()))


if __name__ == "__main__":
    main()
