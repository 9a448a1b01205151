"""Generates tests/golden/camera_side_848x480.json by RUNNING THE REFERENCE's own `setup_camera`
(/root/reference/sim/utils/gs/transform_utils.py:7-31) on the side camera of cfg/env/xarm_gripper.yaml:25-35, on the CPU,
in the authoring container.  The module imports `kornia` and `diff_gaussian_rasterization` at the top and the function
hard-codes `.cuda()`: the two imports are satisfied with placeholder modules (the Camera settings object becomes a plain
namespace that records the keyword arguments) and `Tensor.cuda` is made the identity for the duration of the call, so
every arithmetic operation recorded in the fixture is the reference's own torch code.  The fixture is data only.

Usage (authoring container only):  python tests/golden/make_camera_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/sim/utils/gs/transform_utils.py"
K = [[427.2920227050781, 0.0, 429.9993591308594], [0.0, 426.7926940917969, 242.8115234375], [0.0, 0.0, 1.0]]
C2W = [[0.005258014128948334, 0.6125512321694572, -0.7904133989597472, 0.8830263898083726],
       [0.9999860093046595, -0.0036779908994199082, 0.0038017861441641317, 0.05390846195611962],
       [-0.000578344501100992, -0.7904223303719503, -0.6125620010799026, 0.3976033855145515],
       [0.0, 0.0, 0.0, 1.0]]


def load_reference():
    for name in ("kornia", "diff_gaussian_rasterization"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "diff_gaussian_rasterization":
                m.GaussianRasterizationSettings = lambda **kw: types.SimpleNamespace(**kw)
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_transform_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    w, h = 848, 480
    w2c = np.linalg.inv(np.array(C2W))  # gs_renderer computes w2c = inv(c2w) in float64 numpy
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        cam = ref.setup_camera(w, h, K, w2c, near=0.01, far=100.0, device="cpu")
    finally:
        torch.Tensor.cuda = saved
    out = dict(w=w, h=h, K=K, c2w=C2W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, viewmatrix=[float(x) for x in cam.viewmatrix.reshape(-1)],
               projmatrix=[float(x) for x in cam.projmatrix.reshape(-1)], campos=[float(x) for x in cam.campos], z_threshold=cam.z_threshold,
               scale_modifier=cam.scale_modifier, sh_degree=cam.sh_degree, prefiltered=cam.prefiltered, bg=[float(x) for x in cam.bg])
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera_side_848x480.json"), "w"), indent=1)
    print({k: (v if not isinstance(v, list) or len(v) < 4 else v[:4]) for k, v in out.items()})


if __name__ == "__main__":
    main()
