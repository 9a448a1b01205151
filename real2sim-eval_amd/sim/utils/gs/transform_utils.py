"""``setup_camera`` of the reference (sim/utils/gs/transform_utils.py:7-31): K, w2c -> the 12-field
``GaussianRasterizationSettings`` the rasteriser consumes (row R0 of SURVEY.md §8a).  Only this function of the
reference module is on the hot path; LBS skinning (``interpolate_motions``) is a "next" row."""
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
