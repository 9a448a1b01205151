"""Drop-in operator surface of the reference's ``diff_gaussian_rasterization`` package
(third-party/diff-gaussian-rasterization-w-depth/diff_gaussian_rasterization/__init__.py),
forward only, backed by the hand-written HIP rasteriser of ``libr2s_hip.so`` on MI355X.

Kept identical for callers (sim/renderer/gs_renderer.py:23, sim/utils/gs/transform_utils.py:4):
``GaussianRasterizationSettings`` (12 fields, reference :135-147), ``GaussianRasterizer(raster_settings)``
whose call returns ``(color[3,H,W], radii[P] int32, depth[1,H,W])`` (reference :165-198), the same
argument checks and exception text, and ``rasterize_gaussians`` (reference :17-38).

Out of scope: the backward pass — every render call in the reference runs under
``torch.no_grad`` and depth has no backward; asking for gradients raises ``NotImplementedError``.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from r2s_hip import raster as _raster


class _C:
    """Stand-in for the reference's pybind module ``_C`` (ext.cpp:15-19)."""

    rasterize_gaussians = staticmethod(_raster.rasterize_gaussians)

    @staticmethod
    def mark_visible(positions, viewmatrix, projmatrix):
        # checkFrustum, cuda_rasterizer/rasterizer_impl.cu:54-66: hard-coded z threshold 0.01.
        # Never called by sim/ or experiments/ (SURVEY.md §2.2); a few torch ops are enough.
        vm = viewmatrix.reshape(4, 4)
        z = positions[:, 0] * vm[0, 2] + positions[:, 1] * vm[1, 2] + positions[:, 2] * vm[2, 2] + vm[3, 2]
        return z > 0.01

    @staticmethod
    def rasterize_gaussians_backward(*_args, **_kw):
        raise NotImplementedError("backward pass is out of scope: the simulator renders under torch.no_grad()")


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        # argument order of the C++ entry point, reference :56-76
        args = (
            raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
            raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
            raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.image_height,
            raster_settings.image_width, sh, raster_settings.sh_degree, raster_settings.campos,
            raster_settings.prefiltered, raster_settings.z_threshold,
        )
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth = _C.rasterize_gaussians(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _radii, _depth):
        raise NotImplementedError(
            "diff_gaussian_rasterization (MI355X build) is forward-only; render under torch.no_grad() as "
            "sim/renderer/gs_renderer.py does")


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    z_threshold: float


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)
