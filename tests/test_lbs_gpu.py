"""GPU parity of the HIP skinning kernels: against the reference-generated fixtures (tests/golden/lbs_*.npz, produced by
running the reference's own interpolate_motions) and against the pinned numpy oracle on larger random cases."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "lbs_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_drop_in_interpolate_motions_matches_reference_fixture(path):
    import torch
    from sim.utils.gs.transform_utils import interpolate_motions

    g = np.load(path)
    t = lambda k, dt=None: torch.from_numpy(g[k]).cuda() if dt is None else torch.from_numpy(g[k]).to(dt).cuda()  # noqa: E731
    out, rot, w = interpolate_motions(bones=t("bones"), motions=t("motions"), relations=t("relations", torch.int64), xyz=t("xyz"),
                                      quat=None, weights=t("weights"), weights_indices=t("weights_indices", torch.int64), device="cuda")
    assert rot is None and w.shape == g["weights"].shape
    assert np.abs(out.cpu().numpy() - g["xyz_out"]).max() < 3e-6


def test_batched_envs_large_random_vs_oracle_and_rotations_proper():
    import torch
    from oracle import lbs_oracle
    from r2s_hip.skinning import Skinning

    rng = np.random.default_rng(7)
    N, P, E = 3000, 20000, 3
    bones0 = rng.uniform(-0.08, 0.08, (N, 3)).astype(np.float32)
    rel = lbs_oracle.knn_relations(bones0, 8)
    xyz0 = (bones0[rng.integers(0, N, P)] + rng.normal(0, 0.003, (P, 3))).astype(np.float32)
    w, wi = lbs_oracle.knn_weights(bones0, xyz0, 16)
    bones = np.stack([bones0 + rng.normal(0, 1e-4, bones0.shape).astype(np.float32) for _ in range(E)])
    mot = np.stack([np.cross(rng.normal(0, 0.5, 3), bones0) + rng.normal(0, 0.002, bones0.shape) for _ in range(E)]).astype(np.float32)
    xyz = np.stack([xyz0] * E)
    sk = Skinning(rel, w, wi, device="cuda:0")
    out = sk.interpolate_motions(torch.from_numpy(bones).cuda(), torch.from_numpy(mot).cuda(), torch.from_numpy(xyz).cuda()).cpu().numpy()
    R, flags = sk.debug(E)
    assert int(flags.sum()) == 0
    Rn = R.cpu().numpy().astype(np.float64)
    assert np.allclose(np.linalg.det(Rn), 1.0, atol=1e-5) and np.allclose(Rn @ Rn.transpose(0, 1, 3, 2), np.eye(3), atol=1e-5)
    for e in range(E):
        ref = lbs_oracle.interpolate_motions(bones[e], mot[e], rel, xyz[e], w, wi)
        assert np.abs(out[e] - ref).max() < 3e-6, e


def test_rank_deficient_environment_falls_back_to_identity_like_the_reference():
    import torch
    from oracle import lbs_oracle
    from r2s_hip.skinning import Skinning

    bones = np.zeros((6, 3), np.float32); bones[:, 0] = np.arange(6) * 0.01
    motions = np.tile(np.array([[0.0, 0.01, 0.0]], np.float32), (6, 1)); motions[:, 1] += bones[:, 0]
    rel = np.array([[(i + 1) % 6, (i + 2) % 6] for i in range(6)], np.int32)
    xyz = (bones + 0.001).astype(np.float32)
    w, wi = lbs_oracle.knn_weights(bones, xyz, 3)
    sk = Skinning(rel, w, wi, device="cuda:0")
    out = sk.interpolate_motions(torch.from_numpy(bones).cuda(), torch.from_numpy(motions).cuda(), torch.from_numpy(xyz).cuda()).cpu().numpy()
    ref = lbs_oracle.interpolate_motions(bones, motions, rel, xyz, w, wi)
    _, flags = sk.debug(1)
    assert int(flags[0]) == 1 and np.abs(out - ref).max() < 1e-6
