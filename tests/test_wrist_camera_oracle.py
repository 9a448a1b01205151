"""The numpy restatement of the reference's wrist-camera matrices (oracle/camera_oracle.py) against the fixture the reference's
own render_wrist + setup_camera produced (tests/golden/wrist_camera.npz, make_wrist_camera_golden.py).  CPU only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def scaled_ulps(a, ref):
    """max |a - ref| in units of the float32 spacing at the LARGEST |ref| element of the array: the inverses and products that
    make these matrices carry an absolute error of a few ulps of their largest entries, so the small entries (a camera axis
    component near zero) are not held to their own, much finer, spacing."""
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / np.spacing(np.float32(np.abs(ref).max())))


def test_wrist_camera_oracle_reproduces_the_reference_fixture():
    from oracle.camera_oracle import wrist_camera

    G = np.load(os.path.join(HERE, "golden", "wrist_camera.npz"))
    n = len(G["eef_xyz"])
    assert n >= 10
    for i in range(n):
        view, proj, pos = wrist_camera(G["eef_xyz"][i], G["eef_rot"][i], G["eef2c"], G["K"], int(G["W"]), int(G["H"]), float(G["near"]), float(G["far"]))
        assert np.array_equal(view, G["viewmatrix"][i])              # numpy on both sides: the same LAPACK inverse
        assert scaled_ulps(proj, G["projmatrix"][i]) <= 2            # torch bmm vs numpy matmul
        assert scaled_ulps(pos, G["campos"][i]) <= 2                 # torch.inverse vs numpy.linalg.inv (float32)
    # the rotation block of every view matrix is a rotation: the fixture is a rigid camera
    R = G["viewmatrix"][:, :3, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
