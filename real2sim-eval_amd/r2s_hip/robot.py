"""Python host of include/r2s_robot.h: the robot-link Gaussians of a scan follow their links, for a batch of environments,
written straight into the rasteriser's per-environment Gaussian set.  Replaces ``transform_gs_xarm_gripper`` /
``transform_gs_xarm_pusher`` (sim/utils/robot/robot_pc_transformations.py:12-55, :94-133) and the scene concatenation of
``GSRenderer.update_rendervar`` (sim/renderer/gs_renderer.py:886-921).  Forward kinematics stays with the caller."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, cur_stream

GRIPPER_LINKS = (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16)   # link_id_list of the 18-link gripper arm (:33)
PUSHER_LINKS = (1, 2, 3, 4, 5, 6, 7, 8, 10)                             # of the 11-link pusher arm (:113)

_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if _bound:
        return L
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.r2s_robot_gs_create.restype = C.c_int
    L.r2s_robot_gs_create.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, C.POINTER(vp), vp]
    L.r2s_robot_gs_destroy.restype = None
    L.r2s_robot_gs_destroy.argtypes = [vp]
    L.r2s_robot_gs_transform.restype = C.c_int
    L.r2s_robot_gs_transform.argtypes = [vp, i32, vp, vp, i64, vp, i64, i32, i32, vp]
    L.r2s_robot_gs_debug.restype = C.c_int
    L.r2s_robot_gs_debug.argtypes = [vp, C.POINTER(vp)]
    _bound = True
    return L


class RobotGaussians:
    """``n_links`` robot links, of which ``link_ids`` move Gaussians; ``offsets`` [n_links,4,4] (RobotPcSampler.offsets),
    ``link_pose_base`` [n_links,4,4] (FK of base_qpos); the scan: ``means`` [n,3], ``rotations`` [n,4] (as stored),
    ``total_mask`` [n] link id per Gaussian."""

    def __init__(self, n_links, link_ids, offsets, link_pose_base, means, rotations, total_mask, device="cuda:0"):
        L = _bind()
        self.device = torch.device(device)
        self.n_links = int(n_links)
        listed = np.zeros(self.n_links, np.int32)
        listed[list(link_ids)] = 1
        off = np.ascontiguousarray(np.asarray(offsets, np.float64).reshape(self.n_links, 4, 4))
        base = np.ascontiguousarray(np.asarray(link_pose_base, np.float32).reshape(self.n_links, 4, 4))
        m = np.ascontiguousarray(np.asarray(means, np.float32).reshape(-1, 3))
        q = np.ascontiguousarray(np.asarray(rotations, np.float32).reshape(-1, 4))
        tm = np.ascontiguousarray(np.asarray(total_mask, np.int32).reshape(-1))
        assert len(m) == len(q) == len(tm)
        self.n = int(len(m))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(L.r2s_robot_gs_create(self.n_links, listed.ctypes.data, off.ctypes.data, base.ctypes.data, self.n, m.ctypes.data, q.ctypes.data,
                                        tm.ctypes.data, C.byref(h), cur_stream(self.device)), "r2s_robot_gs_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _bind().r2s_robot_gs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, link_pose: torch.Tensor, means_out: torch.Tensor, rotations_out: torch.Tensor, normalize=True, write_static=True):
        """``link_pose`` [n_env, n_links, 4, 4] (device).  ``means_out`` / ``rotations_out``: [n_env, >= n, 3 / 4] views whose
        first ``n`` rows per environment receive the scan (any environment stride, rows contiguous)."""
        lp = link_pose.to(self.device, torch.float32).contiguous().reshape(-1, self.n_links, 4, 4)
        E = int(lp.shape[0])
        assert means_out.shape[0] == E and rotations_out.shape[0] == E and means_out.shape[1] >= self.n and rotations_out.shape[1] >= self.n
        assert means_out.stride(2) == 1 and means_out.stride(1) == 3 and rotations_out.stride(2) == 1 and rotations_out.stride(1) == 4
        assert means_out.dtype == torch.float32 and rotations_out.dtype == torch.float32
        with torch.cuda.device(self.device):
            check(_bind().r2s_robot_gs_transform(self._h, E, lp.data_ptr(), means_out.data_ptr(), int(means_out.stride(0)) if E > 1 else 0,
                                                 rotations_out.data_ptr(), int(rotations_out.stride(0)) if E > 1 else 0, int(bool(normalize)), int(bool(write_static)),
                                                 cur_stream(self.device)), "r2s_robot_gs_transform")
        self._keep = lp

    def link_records(self, n_env):
        """[n_env, n_links, 16] copy: mat[:3,:4] row-major + quaternion (w, x, y, z) of the last call."""
        from .raster import _memcpy_d2d

        p = C.c_void_p()
        check(_bind().r2s_robot_gs_debug(self._h, C.byref(p)), "r2s_robot_gs_debug")
        t = torch.empty(n_env, self.n_links, 16, dtype=torch.float32, device=self.device)
        _memcpy_d2d(t.data_ptr(), p.value, t.numel() * 4, self.device)
        return t
