"""``setup_camera`` of the reference (sim/utils/gs/transform_utils.py:7-31): K, w2c -> the 12-field
``GaussianRasterizationSettings`` the rasteriser consumes (row R0 of SURVEY.md §8a).  Only this function of the
reference module is on the raster path; LBS skinning (``interpolate_motions``, row f1) is below."""
import weakref

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera


def setup_camera(w, h, k, w2c, near=0.01, far=100.0, bg=[0, 0, 0], z_threshold=0.2, sh_degree=0, device='cuda'):
    fx, fy, cx, cy = k[0][0], k[1][1], k[0][2], k[1][2]
    w2c = torch.as_tensor(w2c).to(device).float()
    cam_center = torch.inverse(w2c)[:3, 3]
    w2c = w2c.unsqueeze(0).transpose(1, 2)
    opengl_proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                                [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                                [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                                [0.0, 0.0, 1.0, 0.0]]).to(device).float().unsqueeze(0).transpose(1, 2)
    full_proj = w2c.bmm(opengl_proj)
    return Camera(
        image_height=h, image_width=w, tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
        bg=torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier=1.0, viewmatrix=w2c.to(device),
        projmatrix=full_proj.to(device), sh_degree=sh_degree, campos=cam_center.to(device), prefiltered=False,
        z_threshold=z_threshold)


# ---- linear-blend skinning (row f1 of SURVEY.md §8f) --------------------------------------------------------------
_SKIN_CACHE = {}


def interpolate_motions(bones, motions, relations, xyz, rot=None, quat=None, weights=None, weights_indices=None, device='cuda',
                        step='n/a'):
    """Drop-in for the reference's ``interpolate_motions`` (sim/utils/gs/transform_utils.py:58-212) as the simulator calls
    it (gs_renderer.py:738-747): returns ``(xyz_transformed, rot, weights)``.  The per-bone Kabsch fit and the blend run
    as two HIP kernels (r2s_skin_interpolate_motions).  With ``quat`` [n_particles, 4] the second return value is the splats'
    rotated quaternions (:197-210, a third kernel); with ``quat=None`` (the simulator) it is ``rot`` passed through."""
    from r2s_hip.skinning import Skinning

    cacheable = weights is not None and isinstance(relations, torch.Tensor)   # weights=None: recomputed from xyz / bones on every call
    if weights is None:  # sparsified weights over the 5 nearest bones, reference :166-174
        dist = torch.norm(xyz[:, None] - bones, dim=-1)
        _, indices = torch.topk(dist, 5, dim=-1, largest=False)
        dist = torch.norm(bones[indices] - xyz[:, None], dim=-1)
        weights = 1 / (dist + 1e-6)
        weights = weights / weights.sum(dim=-1, keepdim=True)
        weights_indices = indices
    assert weights_indices is not None
    assert weights_indices.shape[0] == weights.shape[0] == xyz.shape[0]
    assert weights_indices.shape[1] == weights.shape[1]
    rel_t = relations if isinstance(relations, torch.Tensor) else torch.as_tensor(relations)
    if not cacheable:
        sk = Skinning(rel_t, weights, weights_indices, n_bones=bones.shape[0], device=xyz.device)
        out = sk.interpolate_motions(bones, motions, xyz)
        return out, (sk.rotate_quats(quat) if quat is not None else rot), weights
    # The uploaded topology is cached per (relations, weights, weights_indices) OBJECT: the renderer keeps these three
    # tensors for the lifetime of a scene (gs_renderer.py:195-211).  Identity is checked through weak references and the
    # tensors' version counters, so a freed tensor whose address is reused, or an in-place edit, can never hit a stale entry.
    srcs = (relations if isinstance(relations, torch.Tensor) else None, weights, weights_indices)
    key = tuple(id(t) for t in srcs) + (tuple(rel_t.shape), tuple(weights.shape), int(bones.shape[0]), str(xyz.device))
    ent = _SKIN_CACHE.get(key)
    if ent is not None:
        sk, refs, versions = ent
        if any((r() is not t) if t is not None else False for r, t in zip(refs, srcs)) or versions != tuple(t._version if t is not None else -1 for t in srcs):
            ent = None
    if ent is None:
        if len(_SKIN_CACHE) > 8:
            _SKIN_CACHE.clear()
        sk = Skinning(rel_t, weights, weights_indices, n_bones=bones.shape[0], device=xyz.device)
        _SKIN_CACHE[key] = (sk, tuple(weakref.ref(t) if t is not None else (lambda: None) for t in srcs),
                            tuple(t._version if t is not None else -1 for t in srcs))
    out = sk.interpolate_motions(bones, motions, xyz)
    return out, (sk.rotate_quats(quat) if quat is not None else rot), weights
