"""Drop-in for the asset half of the reference's ``GSProcessor`` (sim/utils/gs/gs_processor.py:15-171): ``load``,
``load_phystwin``, ``save``, ``rotate``, ``translate``, ``scale``, ``crop``, ``apply_mask``, ``merge`` on the same
parameter dictionary (torch float32: means3D [n,3], sh_colors [n,48], log_scales [n,3], unnorm_rotations [n,4] wxyz,
logit_opacities [n,1]).  PLY I/O goes through ``r2s_hip.assets`` (no ``plyfile``); the quaternion algebra that the
reference takes from kornia is done with the numpy helpers there.  The viewer / .splat export helpers are not provided."""
from __future__ import annotations

import numpy as np
import torch

from r2s_hip import assets

_KEYS = ("means3D", "sh_colors", "log_scales", "unnorm_rotations", "logit_opacities")


def _t(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(torch.float32) for k, v in d.items()}


def _quat_to_mat(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))


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

    def rotate(self, params, rot_mat):                                             # :102-120
        rot_mat = np.asarray(rot_mat, np.float32)
        pts = params["means3D"] @ torch.from_numpy(rot_mat).to(params["means3D"]).T
        q = torch.nn.functional.normalize(params["unnorm_rotations"], dim=-1).detach().cpu().numpy().astype(np.float64)
        new_R = rot_mat.astype(np.float64)[None] @ _quat_to_mat(q)
        quats = torch.from_numpy(assets.rot_mats_to_quats(new_R)).to(params["means3D"].device)
        quats = torch.nn.functional.normalize(quats, dim=-1)
        return dict(means3D=pts, sh_colors=params["sh_colors"], log_scales=params["log_scales"], unnorm_rotations=quats,
                    logit_opacities=params["logit_opacities"])

    def translate(self, params, translation):                                      # :122-128
        pts = params["means3D"]
        if isinstance(translation, (list, np.ndarray)):
            translation = torch.tensor(translation, dtype=torch.float32).to(pts.device)
        params["means3D"] = pts + translation
        return params

    def scale(self, params, scale):                                                # :130-137
        pts = params["means3D"]
        if isinstance(scale, (list, np.ndarray)):
            scale = torch.tensor(scale, dtype=torch.float32).to(pts.device)
        params["means3D"] = pts * scale
        params["log_scales"] = torch.log(torch.exp(params["log_scales"]) * scale)
        return params

    def apply_mask(self, params, mask):                                            # :239-247
        return {k: params[k][mask] for k in _KEYS}

    def crop(self, params, bbox, invert=False):                                    # :209-237: axis-aligned [[xmin,xmax],[ymin,ymax],[zmin,zmax]]
        p = params["means3D"]
        b = torch.as_tensor(np.asarray(bbox, np.float32)).to(p.device)
        mask = ((p >= b[:, 0]) & (p <= b[:, 1])).all(dim=-1)
        return self.apply_mask(params, ~mask if invert else mask)

    def merge(self, params_list):                                                  # :290-297
        return {k: torch.cat([p[k] for p in params_list], dim=0) for k in _KEYS}
