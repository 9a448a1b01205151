"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.  numpy restatement of the reference's wrist-camera matrices
(GSRenderer.render_wrist, sim/renderer/gs_renderer.py:966-985, followed by setup_camera, sim/utils/gs/transform_utils.py:7-31):
the checker of the device kernel k_wrist_camera.  PINNED by tests/golden/wrist_camera.npz, which the reference's own code
produced (tests/golden/make_wrist_camera_golden.py)."""
import numpy as np


def wrist_camera(eef_xyz, eef_rot, eef2c, K, w, h, near=0.01, far=100.0):
    """One environment.  Returns viewmatrix [4,4], projmatrix [4,4], campos [3] (float32) like the settings object holds them."""
    e2b = np.eye(4, dtype=np.float32)                                           # torch.eye(4) float32, :975
    e2b[:3, :3] = np.asarray(eef_rot, np.float32)
    e2b[:3, 3] = np.asarray(eef_xyz, np.float32)
    b2eef = np.linalg.inv(e2b)                                                   # float32 in, float32 out, :981
    b2c = np.asarray(eef2c, np.float64) @ b2eef                                  # :983 (float64)
    w2c = b2c @ np.eye(4, dtype=np.float32)                                      # :985
    w2c32 = w2c.astype(np.float32)                                               # torch.tensor(w2c).float(), transform_utils.py:9
    campos = np.linalg.inv(w2c32)[:3, 3].astype(np.float32)                      # :10
    view = np.ascontiguousarray(w2c32.T)                                         # :11
    fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
    proj = np.array([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0], [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                     [0.0, 0.0, far / (far - near), -(far * near) / (far - near)], [0.0, 0.0, 1.0, 0.0]]).astype(np.float32)
    full = (view @ proj.T).astype(np.float32)                                    # bmm, :16
    return view, full, campos
