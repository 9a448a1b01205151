"""GPU parity of the HIP skinning kernels: against the reference-generated fixtures (tests/golden/lbs_*.npz, produced by
running the reference's own interpolate_motions) and against the pinned numpy oracle on larger random cases."""
import glob
import os

import numpy as np
import pytest
from util_parity import close

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
    assert close(out.cpu().numpy(), g["xyz_out"], 3e-6)


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
        assert close(out[e], ref, 3e-6), e


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


def test_drop_in_cache_never_serves_a_stale_topology():
    """the drop-in keeps the uploaded skinning topology per (relations, weights, weights_indices) object.  A new
    scene with the same shapes — whose tensors the caching allocator places at the addresses of the freed ones — must get
    its own topology, an in-place edit of the weights must be seen, and weights=None (recomputed from xyz / bones on every
    call, transform_utils.py:166-174) must never be cached."""
    import torch
    from oracle import lbs_oracle
    from sim.utils.gs.transform_utils import interpolate_motions

    rng = np.random.default_rng(11)
    N, P = 400, 3000
    bones = rng.uniform(-0.05, 0.05, (N, 3)).astype(np.float32)
    rel = lbs_oracle.knn_relations(bones, 8)
    mot = (np.cross(np.array([0.2, -0.1, 0.3]), bones) + rng.normal(0, 0.002, bones.shape)).astype(np.float32)
    tb, tm, tr = torch.from_numpy(bones).cuda(), torch.from_numpy(mot).cuda(), torch.from_numpy(rel.astype(np.int64)).cuda()
    ptrs = set()
    for scene in range(3):                                   # same shapes, different content, old tensors freed in between
        xyz = (bones[rng.integers(0, N, P)] + rng.normal(0, 0.004, (P, 3))).astype(np.float32)
        w, wi = lbs_oracle.knn_weights(bones, xyz, 16)
        tw, twi, tx = torch.from_numpy(w).cuda(), torch.from_numpy(wi.astype(np.int64)).cuda(), torch.from_numpy(xyz).cuda()
        ptrs.add(tw.data_ptr())
        for _ in range(2):                                   # second call hits the cache
            out, _, _ = interpolate_motions(tb, tm, tr, tx, weights=tw, weights_indices=twi)
            assert close(out, lbs_oracle.interpolate_motions(bones, mot, rel, xyz, w, wi), 3e-6), scene
        tw.mul_(0.5); tw[:, 0] += 0.5 * (1 - tw.sum(1) * 2) + 0.5   # in-place edit (rows still sum to 1): version counter changes
        w2 = tw.cpu().numpy()
        out, _, _ = interpolate_motions(tb, tm, tr, tx, weights=tw, weights_indices=twi)
        assert close(out, lbs_oracle.interpolate_motions(bones, mot, rel, xyz, w2, wi), 3e-6), scene
        del tw, twi, tx, out
    # weights=None: the 5-nearest-bone weights are recomputed per call — two different clouds of the same size
    outs = []
    for k in range(2):
        xyz = (bones[rng.integers(0, N, P)] + rng.normal(0, 0.004, (P, 3))).astype(np.float32)
        d = np.linalg.norm(xyz[:, None] - bones[None], axis=-1)
        wi = np.argsort(d, axis=1, kind="stable")[:, :5]
        w = 1.0 / (np.take_along_axis(d, wi, 1) + 1e-6)
        w = (w / w.sum(1, keepdims=True)).astype(np.float32)
        out, _, wret = interpolate_motions(tb, tm, tr, torch.from_numpy(xyz).cuda(), weights=None)
        assert wret.shape == (P, 5)
        assert close(out, lbs_oracle.interpolate_motions(bones, mot, rel, xyz, w, wi), 5e-6), k


def test_quat_branch_rotates_the_splats_like_the_reference_fixture():
    """interpolate_motions(quat=...) through the drop-in: third kernel k_skin_quat, against lbs_quat.npz (made by the reference's
    own function) and against the oracle on a batch."""
    import torch
    from oracle import lbs_oracle
    from sim.utils.gs.transform_utils import interpolate_motions

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbs_quat.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    xyz, rot, _ = interpolate_motions(t(g["bones"]), t(g["motions"]), t(g["relations"].astype(np.int64)), t(g["xyz"]), quat=t(g["quat"]),
                                      weights=t(g["weights"]), weights_indices=t(g["weights_indices"].astype(np.int64)))
    assert close(xyz, g["xyz_out"], 3e-6, what="quat fixture: positions")
    assert close(rot, g["quat_out"], 3e-6, what="quat fixture: rotated quaternions vs reference")
    ref = lbs_oracle.rotate_quats(g["bones"], g["motions"], g["relations"], g["quat"], g["weights"], g["weights_indices"])
    assert close(rot, ref, 3e-6, what="quat fixture: vs oracle")
