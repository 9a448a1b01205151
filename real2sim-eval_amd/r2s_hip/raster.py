"""Python host of the rasteriser C ABI.

``rasterize_gaussians`` plays the role of the reference's torch C++ glue
``RasterizeGaussiansCUDA`` (third-party/diff-gaussian-rasterization-w-depth/rasterize_points.cu:36-117,
bound as ``_C.rasterize_gaussians`` in ext.cpp:16): same argument order, same 7-tuple result.
``RasterBatch`` renders many (environment, camera) frames in one pass.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ALLOC_FN, R2SGaussianSet, R2SRasterDebug, R2SRasterFrame, check, cur_stream

NUM_CHANNELS = 3  # cuda_rasterizer/config.h:14


def _resize_functional(t: torch.Tensor):
    """resizeFunctional, rasterize_points.cu:27-33: grow-on-demand byte tensor as scratch."""

    def cb(_user, nbytes):
        t.resize_(int(nbytes))
        return t.data_ptr()

    return ALLOC_FN(cb)


def _fptr(t: Optional[torch.Tensor]):
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    # the reference takes .contiguous().data<float>() of every argument (rasterize_points.cu:95-114)
    if t.device != device and t.numel() > 0:
        raise ValueError(f"tensor on {t.device}, expected {device} (the reference dereferences raw device pointers)")
    if t.numel() > 0 and t.dtype != torch.float32:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, z_threshold):
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
    if not means3D.is_cuda:
        raise RuntimeError("rasterize_gaussians: means3D must live on a GPU; there is no CPU path")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    out_color = torch.zeros((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    out_depth = torch.zeros((1, H, W), dtype=torch.float32, device=dev)
    geom = torch.empty(0, dtype=torch.uint8, device=dev)
    binning = torch.empty(0, dtype=torch.uint8, device=dev)
    img = torch.empty(0, dtype=torch.uint8, device=dev)
    rendered = 0
    if P != 0:
        M = int(sh.size(1)) if sh.numel() != 0 else 0
        args = [_dev_f32(t, dev) for t in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                            viewmatrix, projmatrix, campos)]
        background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos = args
        cbs = [_resize_functional(geom), _resize_functional(binning), _resize_functional(img)]
        with torch.cuda.device(dev):
            rc = _lib.lib().r2s_raster_forward(
                cbs[0], None, cbs[1], None, cbs[2], None, P, int(degree), M, _fptr(background), W, H, _fptr(means3D),
                _fptr(sh), _fptr(colors), _fptr(opacity), _fptr(scales), float(scale_modifier), _fptr(rotations),
                _fptr(cov3D_precomp), _fptr(viewmatrix), _fptr(projmatrix), _fptr(campos), float(tan_fovx),
                float(tan_fovy), int(bool(prefiltered)), float(z_threshold), _fptr(out_color), _fptr(out_depth),
                _fptr(radii), cur_stream(dev))
        rendered = check(rc, "r2s_raster_forward")
    return rendered, out_color, radii, geom, binning, img, out_depth


class RasterBatch:
    """Batched frames: ``sets`` are Gaussian clouds (one per environment), ``frames`` are camera
    views of a set.  One pass of the pipeline covers all frames (r2s_raster_forward_batch)."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().r2s_raster_ctx_create(C.byref(h)), "r2s_raster_ctx_create")
        self._h = h
        self._keep = []

    def close(self):
        if self._h:
            _lib.lib().r2s_raster_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timing(self, on: bool):
        _lib.lib().r2s_raster_ctx_set_timing(self._h, int(on))

    def set_async(self, on: bool):
        """Sync-free batches (include/r2s_raster.h): no host read of the instance count between scan and emit; ``forward``
        then returns the most recent count the host has seen and ``poll`` reports the last batch."""
        _lib.lib().r2s_raster_ctx_set_async(self._h, int(on))

    def poll(self, wait=False):
        """(still_running, num_rendered, overflow_batches) of the sync-free mode."""
        n, o = C.c_int64(), C.c_int32()
        rc = _lib.lib().r2s_raster_ctx_poll(self._h, int(bool(wait)), C.byref(n), C.byref(o))
        if rc < 0:
            check(rc, "r2s_raster_ctx_poll")
        return bool(rc == 1), int(n.value), int(o.value)

    def set_tile_culling(self, on: bool):
        """Exact-output instance culling (see include/r2s_raster.h); changes the instance count, not the images."""
        _lib.lib().r2s_raster_ctx_set_tile_culling(self._h, int(on))

    def stage_ms(self):
        names = ["preprocess", "scan", "emit", "sort", "ranges", "composite"]
        return {n: float(_lib.lib().r2s_raster_ctx_stage_ms(self._h, i)) for i, n in enumerate(names)}

    def scratch_bytes(self) -> int:
        return int(_lib.lib().r2s_raster_ctx_scratch_bytes(self._h))

    def set_aux(self, final_T: Optional[torch.Tensor], n_contrib: Optional[torch.Tensor]):
        self._aux = (final_T, n_contrib)
        _lib.lib().r2s_raster_ctx_set_aux(self._h, _fptr(final_T), _fptr(n_contrib))

    @staticmethod
    def make_set(means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 scale_modifier=1.0, sh_degree=0):
        keep = [t.contiguous() if t is not None else None for t in (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)]
        means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp = keep
        s = R2SGaussianSet()
        s.P = int(means3D.shape[0]); s.D = int(sh_degree); s.M = int(shs.shape[1]) if shs is not None and shs.numel() else 0
        s.scale_modifier = float(scale_modifier)
        for name, t in (("means3D", means3D), ("shs", shs), ("colors_precomp", colors_precomp), ("opacities", opacities),
                        ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp)):
            setattr(s, name, t.data_ptr() if t is not None and t.numel() else None)
        return s, keep

    def prepare(self, sets: Sequence, frames: Sequence[dict]):
        """The C argument arrays of a (sets, frames) pair that is rendered every step with the same device pointers — built
        once instead of once per call (64 frames x 12 fields of ctypes marshalling are ~0.4 ms of host time per env step)."""
        nS, nF = len(sets), len(frames)
        S = (R2SGaussianSet * max(nS, 1))(*[s for s, _ in sets])
        Fr = (R2SRasterFrame * max(nF, 1))()
        for i, f in enumerate(frames):
            fr = Fr[i]
            fr.set = int(f["set"]); fr.prefiltered = int(bool(f.get("prefiltered", False)))
            fr.tan_fovx = float(f["tanfovx"]); fr.tan_fovy = float(f["tanfovy"]); fr.z_threshold = float(f["z_threshold"])
            fr.viewmatrix = f["viewmatrix"].data_ptr(); fr.projmatrix = f["projmatrix"].data_ptr()
            fr.cam_pos = f["campos"].data_ptr(); fr.background = f["bg"].data_ptr()
            fr.out_color = f["out_color"].data_ptr(); fr.out_depth = f["out_depth"].data_ptr()
            r = f.get("radii")
            fr.radii = r.data_ptr() if r is not None else None
        return (S, nS, Fr, nF, (list(sets), list(frames)))   # keeps the tensors alive

    def forward(self, sets: Sequence, frames: Optional[Sequence[dict]], width: int, height: int, want_counts=False):
        """``sets``: list of (R2SGaussianSet, keepalive) from make_set, ``frames``: dicts with keys set, viewmatrix, projmatrix,
        campos, bg, tanfovx, tanfovy, z_threshold, prefiltered, out_color, out_depth, radii (optional) — or ``sets`` = the result
        of ``prepare`` and ``frames`` = None.  Returns the total instance count (and per-frame counts if asked)."""
        S, nS, Fr, nF, _ = self.prepare(sets, frames) if frames is not None else sets
        counts = (C.c_int64 * max(nF, 1))() if want_counts else None
        with torch.cuda.device(self.device):
            rc = _lib.lib().r2s_raster_forward_batch(self._h, S, nS, Fr, nF, int(width), int(height), counts, cur_stream(self.device))
        n = check(rc, "r2s_raster_forward_batch")
        if want_counts:
            return n, [int(counts[i]) for i in range(nF)]
        return n

    def debug(self):
        """Device intermediates of the last call as torch tensors (copied)."""
        d = R2SRasterDebug()
        check(_lib.lib().r2s_raster_ctx_debug(self._h, C.byref(d)), "r2s_raster_ctx_debug")
        G, L = int(d.total_gaussians), int(d.num_rendered)

        def view(ptr, n, dtype):
            if not ptr or n == 0:
                return torch.empty(0, dtype=dtype)
            nbytes = n * torch.empty(0, dtype=dtype).element_size()
            t = torch.empty(n, dtype=dtype, device=self.device)
            _memcpy_d2d(t.data_ptr(), ptr, nbytes, self.device)
            return t.cpu()

        return dict(
            total_gaussians=G, num_rendered=L,
            depths=view(d.depths, G, torch.float32), radii=view(d.radii, G, torch.int32),
            geom=view(d.geom, G * 12, torch.float32).reshape(G, 12),
            tiles_touched=view(d.tiles_touched, G, torch.int32), point_offsets=view(d.point_offsets, G, torch.int32),
            keys_sorted=view(d.keys_sorted, L, torch.int32), point_list=view(d.point_list, L, torch.int32),
            ranges_ptr=d.ranges,
        )


def _memcpy_d2d(dst_ptr: int, src_ptr: int, nbytes: int, device):
    """Copy raw device memory into a torch tensor (debug taps only)."""
    with torch.cuda.device(device):
        check(_lib.lib().r2s_memcpy_d2d(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes), cur_stream(device)),
              "r2s_memcpy_d2d")
        torch.cuda.synchronize(device)
