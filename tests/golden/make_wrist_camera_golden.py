"""Generates tests/golden/wrist_camera.npz by RUNNING THE REFERENCE's own wrist-camera code
    GSRenderer.render_wrist   (/root/reference/sim/renderer/gs_renderer.py:953-1000: eef2c -> w2c from the gripper pose, :966-985)
    setup_camera              (/root/reference/sim/utils/gs/transform_utils.py:7-31)
on the CPU in the authoring container, for a handful of end-effector poses, with the wrist intrinsics / eef2c of
cfg/env/xarm_gripper.yaml:39-48.  What is NOT the reference's code in this run, and why:
  * kornia.geometry.conversions.quaternion_to_rotation_matrix (third party, absent): restated below from kornia's published
    source (0.7.x, (w, x, y, z) order).  The device kernel takes the ROTATION MATRIX as input (what BaseEnv hands to the
    physics module, phystwin.py:362), so the fixture records the matrix this conversion produced and the conversion itself
    is outside what is pinned;
  * the renderer object is created without __init__ (its constructor needs hydra, SAPIEN, the robot scan ...) and given only
    the attributes render_wrist reads (metadata_wrist, grippers, rendervar_full, cfg.gs.use_shs, device);
  * GaussianRasterizer (the CUDA extension) is a recorder that keeps the GaussianRasterizationSettings it is constructed with;
    Tensor.cuda / Tensor.to(device) are the identity for the duration of the call;
  * open3d / transforms3d / sapien / cv2 / urdfpy / gradio / plyfile imports of the modules: empty placeholder modules (unused here).
Everything else — the 4x4 eef-to-base matrix, its float32 numpy inverse, eef2c @ b2eef, the cast to float32, torch.inverse for
the camera centre, the OpenGL projection and the bmm — is executed from the reference's files.  The fixture is data only.

Usage (authoring container only):  python tests/golden/make_wrist_camera_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
WRIST_K = [[433.2635498046875, 0.0, 425.69775390625], [0.0, 433.2635498046875, 244.70132446289062], [0.0, 0.0, 1.0]]
WRIST_C2EEF = [[-0.00621799798682332, -0.9996882472848673, -0.024181019135736517, 0.070151686668396],
               [0.9999282360076904, -0.0059682438456119995, -0.010387018683749047, -0.006011864222586155],
               [0.01023946, -0.02424387, 0.99965361, 0.03072427], [0.0, 0.0, 0.0, 1.0]]


def kornia_quaternion_to_rotation_matrix(quaternion):
    q = torch.nn.functional.normalize(quaternion, p=2.0, dim=-1, eps=1e-12)
    w, x, y, z = torch.chunk(q, chunks=4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.tensor(1.0)
    m = torch.stack((one - (tyy + tzz), txy - twz, txz + twy, txy + twz, one - (txx + tzz), tyz - twx, txz - twy, tyz + twx, one - (txx + tyy)), dim=-1)
    return m.view(*quaternion.shape[:-1], 3, 3)


class Recorder:
    last = None

    def __init__(self, raster_settings):
        Recorder.last = raster_settings

    def __call__(self, **render_data):
        z = torch.zeros(3, 4, 4)
        return z, None, torch.zeros(1, 4, 4)


def load_reference():
    names = ("open3d", "transforms3d", "sapien", "sapien.core", "kornia", "kornia.geometry", "kornia.geometry.conversions", "cv2", "urdfpy",
             "diff_gaussian_rasterization", "viser", "trimesh", "plyfile", "pytorch3d", "pytorch3d.ops", "pytorch3d.transforms", "gradio")
    for name in names:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["urdfpy"].URDF = object
    sys.modules["plyfile"].PlyData = object
    sys.modules["sapien"].core = sys.modules["sapien.core"]
    k = sys.modules["kornia"]
    k.geometry = sys.modules["kornia.geometry"]
    k.geometry.conversions = sys.modules["kornia.geometry.conversions"]
    k.geometry.conversions.quaternion_to_rotation_matrix = kornia_quaternion_to_rotation_matrix
    d = sys.modules["diff_gaussian_rasterization"]
    d.GaussianRasterizationSettings = lambda **kw: types.SimpleNamespace(**kw)
    d.GaussianRasterizer = Recorder
    sys.path.insert(0, "/root/reference")
    import sim.renderer.gs_renderer as R
    return R


def main():
    R = load_reference()
    rng = np.random.default_rng(11)
    W, H = 848, 480
    ro = object.__new__(R.GSRenderer)
    ro.device = "cpu"
    ro.cfg = types.SimpleNamespace(gs=types.SimpleNamespace(use_shs=False))
    ro.rendervar_full = {"shs": torch.zeros(2, 1, 3), "means3D": torch.zeros(2, 3)}
    ro.online = False
    R.GSRenderer.set_wrist_camera(ro, W, H, np.array(WRIST_K), eef2c=np.linalg.inv(np.array(WRIST_C2EEF)))
    n = 12
    xyz = np.stack([rng.uniform(0.2, 0.6, n), rng.uniform(-0.2, 0.3, n), rng.uniform(0.1, 0.5, n)], 1).astype(np.float32)
    quat = rng.normal(size=(n, 4)).astype(np.float32)
    quat[0] = (0.0, 1.0, 0.0, 0.0)          # tool z axis straight down (rotation by pi about x), the rollout's default pose
    quat[1] = (1.0, 0.0, 0.0, 0.0)          # identity
    rot, view, proj, campos = [], [], [], []
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for i in range(n):
            g = torch.zeros(1, 14)
            g[0, :3] = torch.from_numpy(xyz[i]); g[0, 6:10] = torch.from_numpy(quat[i])
            ro.grippers = g
            R.GSRenderer.render_wrist(ro)
            cam = Recorder.last
            rot.append(kornia_quaternion_to_rotation_matrix(g[:, 6:10])[0].numpy().copy())
            view.append(cam.viewmatrix.numpy().reshape(4, 4).copy()); proj.append(cam.projmatrix.numpy().reshape(4, 4).copy())
            campos.append(cam.campos.numpy().copy())
            meta = dict(tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, z_threshold=cam.z_threshold, sh_degree=cam.sh_degree)
    finally:
        torch.Tensor.cuda = saved_cuda
    out = os.path.join(HERE, "wrist_camera.npz")
    np.savez(out, W=W, H=H, K=np.array(WRIST_K), eef2c=np.linalg.inv(np.array(WRIST_C2EEF)), near=0.01, far=100.0, eef_xyz=xyz, eef_quat=quat,
             eef_rot=np.stack(rot).astype(np.float32), viewmatrix=np.stack(view).astype(np.float32), projmatrix=np.stack(proj).astype(np.float32),
             campos=np.stack(campos).astype(np.float32), tanfovx=meta["tanfovx"], tanfovy=meta["tanfovy"], z_threshold=meta["z_threshold"])
    print(out, np.stack(view)[0], np.stack(campos)[0], meta)


if __name__ == "__main__":
    main()
