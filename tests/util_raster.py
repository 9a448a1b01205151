"""Shared helpers of the raster parity tests (HIP path vs the CPU oracle)."""
import numpy as np


def scene_and_camera(P, W, H, seed, cam="side", sh_coeffs=1, sh_degree=0, z_threshold=0.05, bg=(0.0, 0.0, 0.0)):
    from r2s_hip import synth

    sc = synth.gaussian_scene(P, seed, sh_coeffs=sh_coeffs)
    mk = synth.side_camera if cam == "side" else synth.wrist_camera
    c = mk(W, H, z_threshold=z_threshold, sh_degree=sh_degree, bg=bg)
    return sc, c


def oracle_render(sc, c, f64=False, debug=False, **over):
    import oracle

    kw = dict(shs=sc.get("shs"), scales=sc.get("scales"), rotations=sc.get("rotations"),
              colors_precomp=sc.get("colors_precomp"), cov3D_precomp=sc.get("cov3D_precomp"))
    kw.update(over)
    return oracle.raster_forward(sc["means3D"], sc["opacities"], c["viewmatrix"], c["projmatrix"], c["campos"],
                                 c["tanfovx"], c["tanfovy"], c["image_height"], c["image_width"], c["bg"],
                                 scale_modifier=c["scale_modifier"], sh_degree=c["sh_degree"],
                                 prefiltered=c["prefiltered"], z_threshold=c["z_threshold"], f64=f64, debug=debug, **kw)


def torch_settings(c, device):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    return GaussianRasterizationSettings(
        image_height=c["image_height"], image_width=c["image_width"], tanfovx=c["tanfovx"], tanfovy=c["tanfovy"],
        bg=t(c["bg"]), scale_modifier=c["scale_modifier"], viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]),
        sh_degree=c["sh_degree"], campos=t(c["campos"]), prefiltered=c["prefiltered"], z_threshold=c["z_threshold"])


def hip_render(sc, c, device="cuda:0"):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer

    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    cam = torch_settings(c, device)
    means = t(sc["means3D"])
    with torch.no_grad():
        im, radii, depth = GaussianRasterizer(raster_settings=cam)(
            means3D=means, means2D=torch.zeros_like(means), opacities=t(sc["opacities"]), shs=t(sc.get("shs")),
            colors_precomp=t(sc.get("colors_precomp")), scales=t(sc.get("scales")), rotations=t(sc.get("rotations")),
            cov3D_precomp=t(sc.get("cov3D_precomp")))
    torch.cuda.synchronize()
    return im.cpu().numpy(), radii.cpu().numpy(), depth.cpu().numpy()


def compare_images(color, depth, ref_color, ref_depth, rtol=1e-4, atol=1e-4, what=None):
    """Fraction of pixels whose RGB leaves |d| <= atol + rtol*|ref| and whose median depth differs.  The achieved errors
    (max abs / max rel RGB error, max rel depth error where both sides blended the same median Gaussian, mismatching
    pixel counts) are recorded through util_parity next to the gate they were held to."""
    import inspect
    import os

    from util_parity import record

    d = np.abs(color.astype(np.float64) - ref_color.astype(np.float64))
    bad_rgb = (d > atol + rtol * np.abs(ref_color)).any(0)
    dd = np.abs(depth.astype(np.float64) - ref_depth.astype(np.float64))
    bad_depth = (dd > rtol * np.abs(ref_depth))[0]
    out = dict(frac_rgb=float(bad_rgb.mean()), frac_depth=float(bad_depth.mean()), max_rgb=float(d.max()),
               n_rgb=int(bad_rgb.sum()), n_depth=int(bad_depth.sum()))
    lit = np.abs(ref_color) > 1e-2                       # relative error where the reference channel is not ~black
    same = ~bad_depth                                    # pixels whose median depth comes from the same Gaussian
    if what is None:
        fr = inspect.stack()[1]
        what = f"{os.path.basename(fr.filename)}:{fr.lineno}"
    record(what, max_abs_rgb_err=float(d.max()), max_rel_rgb_err=float((d[lit] / np.abs(ref_color[lit])).max()) if lit.any() else 0.0,
           max_rel_depth_err_same_median=float((dd[0][same] / np.abs(ref_depth[0][same])).max()) if same.any() else 0.0,
           mismatching_rgb_pixels=int(bad_rgb.sum()), median_depth_mismatch_pixels=int(bad_depth.sum()), pixels=int(bad_rgb.size),
           gate_rtol=float(rtol), gate_atol=float(atol), tol=1e-4)
    return out
