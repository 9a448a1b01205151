"""Task-success predicates of the reference's evaluation scripts on the device (include/r2s_metrics.h), batched over
environments, plus the episode rule they share (success = predicate true in >= 30 of the frames looked at)."""
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
        vp, i32 = C.c_void_p, C.c_int32
        L.r2s_metric_plane_crossings.restype = C.c_int
        L.r2s_metric_plane_crossings.argtypes = [i32, i32, vp, i32, vp, vp, vp, C.c_double, vp, vp]
        L.r2s_metric_mse.restype = C.c_int
        L.r2s_metric_mse.argtypes = [i32, i32, vp, vp, vp, vp]
        L.r2s_metric_points_in_obb.restype = C.c_int
        L.r2s_metric_points_in_obb.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp]
        _bound = True
    return L


def _x(x):
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] == 3
    return x.contiguous()


def _d3(a, n=3):
    return np.ascontiguousarray(np.asarray(a, np.float64).reshape(-1)[:n])


# calculate_success_rope.py:150-160: the clip the rope is routed through
ROPE_CLIP_MIN = np.array([0.62 - 0.035 / 2, 0.05 - 0.035 / 2, 0.0])
ROPE_CLIP_MAX = np.array([0.62 + 0.035 / 2, 0.05 + 0.035 / 2, 0.03])


def plane_crossings(x: torch.Tensor, springs: torch.Tensor, bbox_min, bbox_max, eps: float = 1e-12) -> torch.Tensor:
    """x [n_env, N, 3] (cuda float32), springs [S, 2] (cuda int32) -> int32 [n_env, 2] = (y_min_count, y_max_count)."""
    x = _x(x)
    springs = springs.to(x.device, torch.int32).contiguous()
    lo, hi = _d3(bbox_min), _d3(bbox_max)
    out = torch.empty(x.shape[0], 2, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        check(_bind().r2s_metric_plane_crossings(x.shape[0], x.shape[1], x.data_ptr(), springs.shape[0], springs.data_ptr(), lo.ctypes.data, hi.ctypes.data,
                                                 float(eps), out.data_ptr(), cur_stream(x.device)), "r2s_metric_plane_crossings")
    return out


def rope_routed(x, springs) -> torch.Tensor:
    """is_rope_success (calculate_success_rope.py:139-168) per environment -> bool [n_env]."""
    c = plane_crossings(x, springs, ROPE_CLIP_MIN, ROPE_CLIP_MAX)
    return (c[:, 0] >= 100) & (c[:, 1] >= 100)


def mse_to_target(x: torch.Tensor, x_target: torch.Tensor) -> torch.Tensor:
    """((x - x_target)**2).sum(1).mean() per environment -> float64 [n_env]; push-T succeeds below 0.002."""
    x = _x(x)
    t = x_target.to(x.device, torch.float32).contiguous()
    assert t.shape == x.shape[1:]
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(_bind().r2s_metric_mse(x.shape[0], x.shape[1], x.data_ptr(), t.data_ptr(), out.data_ptr(), cur_stream(x.device)), "r2s_metric_mse")
    return out


def pusht_success(x, x_target) -> torch.Tensor:
    return mse_to_target(x, x_target) < 0.002


def points_in_obb(x: torch.Tensor, center, R, half_extent) -> torch.Tensor:
    """Particles inside an oriented box (columns of R = box axes) -> int32 [n_env]; the sloth task counts >= 3050 inside
    the box obstacle's OBB scaled by 1.05 (calculate_success_sloth.py:156-168)."""
    x = _x(x)
    c, r, h = _d3(center), _d3(R, 9), _d3(half_extent)
    out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        check(_bind().r2s_metric_points_in_obb(x.shape[0], x.shape[1], x.data_ptr(), c.ctypes.data, r.ctypes.data, h.ctypes.data, out.data_ptr(),
                                               cur_stream(x.device)), "r2s_metric_points_in_obb")
    return out


class EpisodeSuccess:
    """The episode rule of all three scripts: the predicate must hold in at least ``need`` (30) of the evaluated frames
    (the last 100 of an episode).  Counts stay on the device until ``result`` is read."""

    def __init__(self, n_env, device, need=30):
        self.count = torch.zeros(n_env, dtype=torch.int32, device=device)
        self.need = need

    def update(self, flags: torch.Tensor):
        self.count += flags.to(torch.int32)

    def result(self) -> torch.Tensor:
        return self.count >= self.need
