"""oracle/robot_oracle.py against the fixtures the reference's own transform_gs_xarm_gripper / transform_gs_xarm_pusher produced
(tests/golden/robot_gs_*.npz <- tests/golden/make_robot_gs_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import robot_oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "robot_gs_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_oracle_reproduces_the_reference_transform(path):
    g = np.load(path)
    ids = ro.GRIPPER_LINKS if "gripper" in path else ro.PUSHER_LINKS
    assert len(g["offsets"]) == int(g["n_links"]) == (18 if "gripper" in path else 11)
    for c in range(len(g["link_pose"])):
        m, q = ro.transform_gs(g["means"], g["quats"], g["total_mask"], ids, g["link_pose"][c], g["link_pose_base"], g["offsets"])
        assert np.abs(m - g["new_means"][c]).max() < 2e-6
        assert np.abs(q - g["new_quats"][c]).max() < 2e-6
        static = ~np.isin(g["total_mask"], ids)
        assert static.sum() > 100 and np.array_equal(m[static], g["means"][static])       # unlisted link ids and -1 stay put
        assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-5


def test_quaternion_of_link_matrix_is_the_rotation():
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        th = rng.uniform(-np.pi, np.pi)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32)
        w, x, y, z = ro.rotation_matrix_to_quaternion(R).astype(np.float64)
        R2 = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.abs(R2 - R).max() < 5e-6
