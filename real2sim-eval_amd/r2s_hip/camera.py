"""Python host of include/r2s_camera.h: the wrist camera of every environment from its end-effector pose, on the device
(reference: GSRenderer.render_wrist, sim/renderer/gs_renderer.py:953-1000, + setup_camera, sim/utils/gs/transform_utils.py:7-31)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, cur_stream

_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp = C.c_void_p
        L.r2s_wrist_camera.restype = C.c_int
        L.r2s_wrist_camera.argtypes = [C.c_int32, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_double, C.c_double, vp, vp, vp, vp]
        L.r2s_rot_to_quat.restype = C.c_int
        L.r2s_rot_to_quat.argtypes = [C.c_int32, vp, vp, vp]
        _bound = True
    return L


def rot_to_quat(rot: torch.Tensor) -> torch.Tensor:
    """[n,3,3] device rotation matrices -> [n,4] (w, x, y, z): r2s_rot_to_quat, one launch (r2s_camera.h)."""
    r = rot.to(torch.float32).contiguous().reshape(-1, 9)
    q = torch.empty(r.shape[0], 4, dtype=torch.float32, device=r.device)
    with torch.cuda.device(r.device):
        check(_bind().r2s_rot_to_quat(r.shape[0], r.data_ptr(), q.data_ptr(), cur_stream(r.device)), "r2s_rot_to_quat")
    return q


class WristCamera:
    """``set_wrist_camera(w, h, intr, eef2c, near, far)`` of the reference (gs_renderer.py:181-193) for ``n_env`` environments:
    ``update(eef_xyz, eef_rot)`` rewrites ``viewmatrix`` [n_env,1,4,4], ``projmatrix`` [n_env,1,4,4] and ``campos`` [n_env,3]
    IN PLACE (the rasteriser's prepared frames keep pointing at them).  ``tanfovx`` / ``tanfovy`` are the constant fields."""

    def __init__(self, n_env, w, h, intr, eef2c, near=0.01, far=100.0, device="cuda:0"):
        self.device = torch.device(device)
        self.n_env, self.w, self.h, self.near, self.far = int(n_env), int(w), int(h), float(near), float(far)
        self.K = np.ascontiguousarray(np.asarray(intr, np.float64).reshape(3, 3))
        self.eef2c = np.ascontiguousarray(np.asarray(eef2c, np.float64).reshape(4, 4))
        self.tanfovx = float(w / (2 * self.K[0, 0]))
        self.tanfovy = float(h / (2 * self.K[1, 1]))
        self.viewmatrix = torch.zeros(n_env, 1, 4, 4, dtype=torch.float32, device=self.device)
        self.projmatrix = torch.zeros(n_env, 1, 4, 4, dtype=torch.float32, device=self.device)
        self.campos = torch.zeros(n_env, 3, dtype=torch.float32, device=self.device)

    def update(self, eef_xyz: torch.Tensor, eef_rot: torch.Tensor):
        x = eef_xyz.to(self.device, torch.float32).contiguous().reshape(self.n_env, 3)
        r = eef_rot.to(self.device, torch.float32).contiguous().reshape(self.n_env, 3, 3)
        with torch.cuda.device(self.device):
            check(_bind().r2s_wrist_camera(self.n_env, x.data_ptr(), r.data_ptr(), self.eef2c.ctypes.data, self.K.ctypes.data, self.w, self.h,
                                           self.near, self.far, self.viewmatrix.data_ptr(), self.projmatrix.data_ptr(), self.campos.data_ptr(),
                                           cur_stream(self.device)), "r2s_wrist_camera")
        self._keep = (x, r)
        return self.viewmatrix, self.projmatrix, self.campos
