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


def compare_images(color, depth, ref_color, ref_depth, rtol=1e-4, atol=1e-5, what=None, fragile=None):
    """The image gate (BASELINE.json "RGB/depth within 1e-4 rel").  A pixel's RGB passes when every channel has
    |d| <= atol + rtol |ref| (atol 1e-5 since round 5 — one 8-bit level is 4e-3; it only keeps channels that are black in the
    reference from failing a purely relative test on 1e-7 of rounding); its median depth when |d| <= rtol |ref|.  Pixels that do not pass are CLASSIFIED with the oracle's
    ``fragile`` mask (oracle.raster_forward(fragile=True): a per-pixel decision — alpha < 1/255, power > 0, test_T < 1e-4, the
    median-depth crossing T > 0.5 && test_T < 0.5 — sat within a relative 5e-5 of flipping on that pixel):
      threshold flips        RGB mismatch on a pixel with fragile bit 0 — `v_exp_f32` vs `expf` may decide the other way; the colour
                             then differs by ~alpha T rgb ~ 1/255.  Counted and reported on their own (n_flip_rgb), allowed up to
                             1e-4 of the frame's pixels;
      median-depth crossings depth mismatch on a pixel with bit 1 (or bit 0: a flip moves T) — n_flip_depth, same allowance
                             (the one SURVEY.md §8d names);
      hard mismatches        anything else: an RGB / depth error that NO near-threshold decision explains — n_hard_rgb /
                             n_hard_depth, must be 0 ("rgb within 1e-4" is claimed for every pixel that is not a threshold flip).
    Without a mask (callers that have no oracle intermediates) every mismatch is reported as unclassified.  Returns the counts and
    fractions; the achieved errors are recorded through util_parity next to the gate they were held to."""
    import inspect
    import os

    from util_parity import record

    d = np.abs(color.astype(np.float64) - ref_color.astype(np.float64))
    bad_rgb = (d > atol + rtol * np.abs(ref_color)).any(0)
    dd = np.abs(depth.astype(np.float64) - ref_depth.astype(np.float64))
    bad_depth = (dd > rtol * np.abs(ref_depth))[0]
    out = dict(frac_rgb=float(bad_rgb.mean()), frac_depth=float(bad_depth.mean()), max_rgb=float(d.max()),
               n_rgb=int(bad_rgb.sum()), n_depth=int(bad_depth.sum()))
    if fragile is not None:
        fr0, fr1 = (fragile & 1) != 0, (fragile & 3) != 0
        out.update(n_flip_rgb=int((bad_rgb & fr0).sum()), n_hard_rgb=int((bad_rgb & ~fr0).sum()),
                   n_flip_depth=int((bad_depth & fr1).sum()), n_hard_depth=int((bad_depth & ~fr1).sum()), n_fragile_pixels=int(fr1.sum()))
        calm = (~fr1)[None] & (np.abs(ref_color) > 1e-2)          # lit channels of pixels without any near-threshold decision
        out["max_rel_rgb_non_fragile"] = float((d[calm] / np.abs(ref_color[calm])).max()) if calm.any() else 0.0
    lit = np.abs(ref_color) > 1e-2                       # relative error where the reference channel is not ~black
    same = ~bad_depth                                    # pixels whose median depth comes from the same Gaussian
    if what is None:
        fr = inspect.stack()[1]
        what = f"{os.path.basename(fr.filename)}:{fr.lineno}"
    extra = {k: out[k] for k in ("n_flip_rgb", "n_hard_rgb", "n_flip_depth", "n_hard_depth", "n_fragile_pixels", "max_rel_rgb_non_fragile") if k in out}
    record(what, max_abs_rgb_err=float(d.max()), max_rel_rgb_err=float((d[lit] / np.abs(ref_color[lit])).max()) if lit.any() else 0.0,
           max_rel_depth_err_same_median=float((dd[0][same] / np.abs(ref_depth[0][same])).max()) if same.any() else 0.0,
           mismatching_rgb_pixels=int(bad_rgb.sum()), median_depth_mismatch_pixels=int(bad_depth.sum()), pixels=int(bad_rgb.size),
           gate_rtol=float(rtol), gate_atol=float(atol), tol=1e-4, **extra)
    return out


def assert_image_gate(r, pixels, where=None):
    """The gate itself, on a classified comparison (compare_images(..., fragile=mask)): NO pixel may differ without a near-threshold
    decision explaining it; threshold flips and median-depth crossings may touch at most 1e-4 of the frame's pixels each."""
    assert r["n_hard_rgb"] == 0 and r["n_hard_depth"] == 0, (where, r)
    assert r["n_flip_rgb"] <= 1e-4 * pixels and r["n_flip_depth"] <= 1e-4 * pixels, (where, r)
