"""Drop-in for the FILE half of the reference's ``GSProcessor`` (sim/utils/gs/gs_processor.py:15-171): ``load``, ``load_phystwin``,
``save`` and ``apply_mask`` on the same parameter dictionary (torch float32: means3D [n,3], sh_colors [n,48], log_scales [n,3],
unnorm_rotations [n,4] wxyz, logit_opacities [n,1]) — what ``GSRenderer.load_scaniverse`` calls.  PLY I/O goes through
``r2s_hip.assets`` (no ``plyfile``).  The scan-editing helpers (rotate / translate / scale / crop / merge), the viewer and the
.splat export are out of scope (only the PLY layout matters to the hot paths): they raise NotImplementedError
with a pointer to the reference."""
from __future__ import annotations

import numpy as np
import torch

from r2s_hip import assets

_KEYS = ("means3D", "sh_colors", "log_scales", "unnorm_rotations", "logit_opacities")


def _t(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float32) for k, v in d.items()}


class GSProcessor:
    def load(self, in_dir, rot_x_minus90=False):                                   # :59-100
        return _t(assets.load_gaussians_ply(in_dir, rot_x_minus90=rot_x_minus90))

    def load_phystwin(self, path, max_sh_degrees=3):                              # :19-57: isotropic scale column repeated 3x
        v = assets.read_ply_vertices(path)
        names = v.dtype.names
        rest = sorted([n for n in names if n.startswith("f_rest_")], key=lambda n: int(n.split("_")[-1]))
        assert len(rest) == 3 * (max_sh_degrees + 1) ** 2 - 3
        feats = np.zeros((len(v), len(rest) + 3))
        for k in range(3):
            feats[:, k] = v[f"f_dc_{k}"]
        for k, n in enumerate(rest):
            feats[:, k] = v[n]                                                      # the reference overwrites the DC columns the same way (:34-35)
        scales = np.stack([v[n] for n in sorted([n for n in names if n.startswith("scale_")], key=lambda n: int(n.split("_")[-1]))], -1)
        rots = np.stack([v[n] for n in sorted([n for n in names if n.startswith("rot")], key=lambda n: int(n.split("_")[-1]))], -1)
        out = _t(dict(means3D=np.stack([v["x"], v["y"], v["z"]], -1), sh_colors=feats, log_scales=scales, unnorm_rotations=rots,
                      logit_opacities=np.asarray(v["opacity"])[:, None]))
        out["log_scales"] = out["log_scales"].repeat(1, 3)
        return out

    def save(self, params, save_dir):                                              # :139-171
        assets.save_gaussians_ply({k: params[k].detach().cpu().numpy() for k in _KEYS}, save_dir)

    def apply_mask(self, params, mask):                                            # :239-247
        return {k: params[k][mask] for k in _KEYS}

    # The scan-EDITING half of the reference class (:102-137, :173-237, :249-330) is out of scope (offline asset
    # preparation, not on the evaluation path).  The names exist so that a caller finds out at the call, with a pointer, instead of
    # through an AttributeError.
    def _out_of_scope(self, name):
        raise NotImplementedError(f"GSProcessor.{name}: scan editing is not part of this drop-in (file half only: load / load_phystwin / save / "
                                  f"apply_mask); use the reference's sim/utils/gs/gs_processor.py for offline asset preparation")

    def rotate(self, params, rot_mat):
        self._out_of_scope("rotate")

    def translate(self, params, translation):
        self._out_of_scope("translate")

    def scale(self, params, scale):
        self._out_of_scope("scale")

    def crop(self, params, bbox, invert=False):
        self._out_of_scope("crop")

    def merge(self, params_list):
        self._out_of_scope("merge")

    def save_to_splat(self, params, save_dir, center=True, rotate=True):
        self._out_of_scope("save_to_splat")

    def visualize_gs(self, gs_name_list, transform=False, merged=False, axis_on=False):
        self._out_of_scope("visualize_gs")

    def add_axis(self, params):
        self._out_of_scope("add_axis")
