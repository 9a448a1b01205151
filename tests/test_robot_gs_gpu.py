"""GPU parity of the robot-link Gaussian kernels (include/r2s_robot.h) against the reference-generated fixtures and, batched over
environments with strided outputs, against the pinned oracle."""
import glob
import os

import numpy as np
import pytest

from util_parity import close

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "robot_gs_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_kernels_reproduce_the_reference_transform_all_configurations_as_one_batch(path):
    import torch
    from r2s_hip.robot import GRIPPER_LINKS, PUSHER_LINKS, RobotGaussians

    g = np.load(path)
    ids = GRIPPER_LINKS if "gripper" in path else PUSHER_LINKS
    rg = RobotGaussians(int(g["n_links"]), ids, g["offsets"], g["link_pose_base"], g["means"], g["quats"], g["total_mask"])
    E, n = len(g["link_pose"]), len(g["means"])
    # outputs are views into a larger per-env Gaussian set (what the rasteriser reads): rows [5, 5 + n) of [E, n + 9]
    M = torch.full((E, n + 9, 3), 7.0, device="cuda"); Q = torch.full((E, n + 9, 4), 7.0, device="cuda")
    rg.transform(torch.from_numpy(g["link_pose"]).cuda(), M[:, 5:], Q[:, 5:], normalize=False)
    torch.cuda.synchronize()
    assert close(M[:, 5:5 + n], g["new_means"], 2e-6, what="means vs reference transform_gs_xarm_*")
    assert close(Q[:, 5:5 + n], g["new_quats"], 2e-6, what="rotations vs reference transform_gs_xarm_*")
    assert float((M[:, :5] - 7).abs().max()) == 0 and float((M[:, 5 + n:] - 7).abs().max()) == 0 and float((Q[:, :5] - 7).abs().max()) == 0
    rg.transform(torch.from_numpy(g["link_pose"]).cuda(), M[:, 5:], Q[:, 5:], normalize=True)
    ref = g["new_quats"] / np.maximum(np.linalg.norm(g["new_quats"], axis=-1, keepdims=True), 1e-12)
    assert close(Q[:, 5:5 + n], ref, 2e-6, what="rotations after the renderer's final normalisation")
    rec = rg.link_records(E).cpu().numpy()
    from oracle import robot_oracle as ro
    for e in range(E):
        mats, quats = ro.link_matrices(g["link_pose"][e], g["link_pose_base"], g["offsets"])
        for l in ids:
            assert np.abs(rec[e, l, :12].reshape(3, 4) - mats[l][:3]).max() < 2e-6 and np.abs(rec[e, l, 12:] - quats[l]).max() < 2e-6, (e, l)


def test_large_scan_many_envs_vs_oracle():
    import torch
    from oracle import robot_oracle as ro
    from r2s_hip.robot import GRIPPER_LINKS, RobotGaussians

    g = np.load(os.path.join(HERE, "golden", "robot_gs_gripper.npz"))
    rng = np.random.default_rng(5)
    n, E = 60000, 8
    means = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
    quats = rng.normal(size=(n, 4)).astype(np.float32)
    mask = rng.integers(-1, 18, n).astype(np.int32)
    pose = np.stack([g["link_pose"][rng.integers(0, len(g["link_pose"]))][rng.permutation(18)] for _ in range(E)])   # any rigid matrices do
    rg = RobotGaussians(18, GRIPPER_LINKS, g["offsets"], g["link_pose_base"], means, quats, mask)
    M = torch.empty(E, n, 3, device="cuda"); Q = torch.empty(E, n, 4, device="cuda")
    rg.transform(torch.from_numpy(pose).cuda(), M, Q, normalize=True)
    for e in (0, E - 1):
        m, q = ro.transform_gs(means, quats, mask, GRIPPER_LINKS, pose[e], g["link_pose_base"], g["offsets"], final_normalize=True)
        assert close(M[e], m, 3e-6, what="means, 60k-Gaussian scan") and close(Q[e], q, 3e-6, what="rotations, 60k-Gaussian scan")
