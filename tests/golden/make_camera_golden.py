"""Generates tests/golden/camera_side_848x480.json: the reference's setup_camera arithmetic
(sim/utils/gs/transform_utils.py:7-31) evaluated with torch float32 CPU ops on the side camera of
cfg/env/xarm_gripper.yaml:25-35.  The reference function itself cannot be imported here (it needs kornia and
diff_gaussian_rasterization._C at import time, and hard-codes .cuda()), so this script re-evaluates its
formulas line by line with torch — same dtype, same op order — and stores inputs + outputs as data."""
import json
import os

import numpy as np
import torch

K = [[427.2920227050781, 0.0, 429.9993591308594], [0.0, 426.7926940917969, 242.8115234375], [0.0, 0.0, 1.0]]
C2W = [[0.005258014128948334, 0.6125512321694572, -0.7904133989597472, 0.8830263898083726],
       [0.9999860093046595, -0.0036779908994199082, 0.0038017861441641317, 0.05390846195611962],
       [-0.000578344501100992, -0.7904223303719503, -0.6125620010799026, 0.3976033855145515],
       [0.0, 0.0, 0.0, 1.0]]
w, h, near, far = 848, 480, 0.01, 100.0
w2c = np.linalg.inv(np.array(C2W))  # gs_renderer computes w2c = inv(c2w) in float64 numpy
fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
t = torch.tensor(w2c).float()
cam_center = torch.inverse(t)[:3, 3]
t = t.unsqueeze(0).transpose(1, 2)
proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0], [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                     [0.0, 0.0, far / (far - near), -(far * near) / (far - near)], [0.0, 0.0, 1.0, 0.0]]).float().unsqueeze(0).transpose(1, 2)
full = t.bmm(proj)
out = dict(w=w, h=h, K=K, c2w=C2W, tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
           viewmatrix=[float(x) for x in t.reshape(-1)], projmatrix=[float(x) for x in full.reshape(-1)],
           campos=[float(x) for x in cam_center])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera_side_848x480.json"), "w"), indent=1)
