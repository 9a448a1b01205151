"""Python host of the skinning C ABI (include/r2s_skinning.h): Gaussians follow the PhysTwin particles.

`Skinning` holds the one-time topology (relations among bones, per-point bone weights) and applies the per-env-step
update for a batch of environments; `interpolate_motions` in ``sim/utils/gs/transform_utils.py`` wraps it with the
reference's signature."""
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
        L.r2s_skin_create.restype = C.c_int
        L.r2s_skin_create.argtypes = [i32, i32, vp, i32, i32, vp, vp, C.POINTER(vp), vp]
        L.r2s_skin_destroy.restype = None
        L.r2s_skin_destroy.argtypes = [vp]
        L.r2s_skin_interpolate_motions.restype = C.c_int
        L.r2s_skin_interpolate_motions.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.r2s_skin_interpolate_motions_strided.restype = C.c_int
        L.r2s_skin_interpolate_motions_strided.argtypes = [vp, i32, vp, vp, vp, C.c_int64, vp, C.c_int64, vp]
        L.r2s_skin_debug.restype = C.c_int
        L.r2s_skin_debug.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.r2s_skin_rotate_quats.restype = C.c_int
        L.r2s_skin_rotate_quats.argtypes = [vp, i32, vp, C.c_int64, vp, C.c_int64, vp]
        _bound = True
    return L


def _np(a, dtype):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


class Skinning:
    def __init__(self, relations, weights, weights_indices, n_bones=None, device="cuda:0"):
        L = _bind()
        self.device = torch.device(device)
        rel = _np(relations, np.int32)
        w = _np(weights, np.float32)
        wi = _np(weights_indices, np.int32)
        self.n_bones = int(n_bones if n_bones is not None else rel.shape[0])
        assert rel.shape[0] == self.n_bones and w.shape == wi.shape
        self.k_rel, self.n_points, self.k_wgt = int(rel.shape[1]), int(w.shape[0]), int(w.shape[1])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(L.r2s_skin_create(self.n_bones, self.k_rel, rel.ctypes.data, self.n_points, self.k_wgt, w.ctypes.data, wi.ctypes.data,
                                    C.byref(h), cur_stream(self.device)), "r2s_skin_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _bind().r2s_skin_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def interpolate_motions(self, bones, motions, xyz, out=None):
        """bones, motions: [n_env, n_bones, 3] (or [n_bones, 3]); xyz: [n_env, n_points, 3] -> transformed xyz (same shape).
        ``xyz`` / ``out`` may be views with an environment stride (rows contiguous), e.g. the first ``n_points`` rows of the
        rasteriser's per-environment Gaussian set — ``out`` may be ``xyz`` itself (in place)."""
        single = bones.dim() == 2
        b = bones.to(self.device, torch.float32).contiguous().reshape(-1, self.n_bones, 3)
        m = motions.to(self.device, torch.float32).contiguous().reshape(-1, self.n_bones, 3)
        x = xyz.to(self.device, torch.float32)
        x = x.reshape(-1, self.n_points, 3) if x.dim() == 2 else x
        if not (x.stride(2) == 1 and x.stride(1) == 3):
            x = x.contiguous()
        E = b.shape[0]
        assert m.shape[0] == E and x.shape[0] == E and x.shape[1] == self.n_points
        if out is None:
            out = torch.empty(E, self.n_points, 3, dtype=torch.float32, device=self.device)
        assert out.shape == x.shape and out.stride(2) == 1 and out.stride(1) == 3 and out.dtype == torch.float32
        with torch.cuda.device(self.device):
            check(_bind().r2s_skin_interpolate_motions_strided(self._h, E, b.data_ptr(), m.data_ptr(), x.data_ptr(), int(x.stride(0)), out.data_ptr(),
                                                               int(out.stride(0)), cur_stream(self.device)), "r2s_skin_interpolate_motions_strided")
        return out[0] if single else out

    def rotate_quats(self, quat, out=None):
        """``interpolate_motions(quat=...)`` (transform_utils.py:197-210) with the bone rotations of the LAST ``interpolate_motions``
        call: quat [n_env, n_points, 4] (or [n_points, 4]), (w, x, y, z) -> blended bone rotation (x) quat, same shape."""
        single = quat.dim() == 2
        q = quat.to(self.device, torch.float32)
        q = q.reshape(-1, self.n_points, 4) if single else q
        if not (q.stride(2) == 1 and q.stride(1) == 4):
            q = q.contiguous()
        E = q.shape[0]
        if out is None:
            out = torch.empty(E, self.n_points, 4, dtype=torch.float32, device=self.device)
        assert out.shape == q.shape and out.stride(2) == 1 and out.stride(1) == 4 and out.dtype == torch.float32
        with torch.cuda.device(self.device):
            check(_bind().r2s_skin_rotate_quats(self._h, E, q.data_ptr(), int(q.stride(0)), out.data_ptr(), int(out.stride(0)), cur_stream(self.device)),
                  "r2s_skin_rotate_quats")
        return out[0] if single else out

    def debug(self, n_env=1):
        """(rotations [n_env, n_bones, 3, 3], identity flags [n_env]) of the last call."""
        from .raster import _memcpy_d2d

        r, f = C.c_void_p(), C.c_void_p()
        check(_bind().r2s_skin_debug(self._h, C.byref(r), C.byref(f)), "r2s_skin_debug")
        R = torch.empty(n_env, self.n_bones, 3, 3, dtype=torch.float32, device=self.device)
        F = torch.empty(n_env, dtype=torch.int32, device=self.device)
        _memcpy_d2d(R.data_ptr(), r.value, R.numel() * 4, self.device)
        _memcpy_d2d(F.data_ptr(), f.value, F.numel() * 4, self.device)
        return R, F


# ---- one-time topology, as GSRenderer builds it (gs_renderer.py:195-211): host side, runs once per scene ------------
def knn_relations(bones, k=8):
    """k nearest other bones of every bone (KD-tree, self excluded) -> int32 [n_bones, k]."""
    from scipy.spatial import cKDTree

    b = np.asarray(bones, np.float64)
    _, idx = cKDTree(b).query(b, k=k + 1)
    return np.ascontiguousarray(idx[:, 1:], dtype=np.int32)


def knn_weights(bones, pts, k=16):
    """Inverse-distance weights over the k nearest bones of every point -> (float32 [P, k], int32 [P, k])."""
    from scipy.spatial import cKDTree

    b = np.asarray(bones, np.float32)
    p = np.asarray(pts, np.float32)
    _, idx = cKDTree(b.astype(np.float64)).query(p.astype(np.float64), k=k)
    idx = idx.reshape(len(p), k)
    dist = np.linalg.norm(b[idx] - p[:, None], axis=-1).astype(np.float32)
    w = (1.0 / (dist + np.float32(1e-6))).astype(np.float32)
    w = w / w.sum(-1, keepdims=True)
    return np.ascontiguousarray(w, np.float32), np.ascontiguousarray(idx, np.int32)
