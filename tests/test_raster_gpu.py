"""GPU parity of the HIP rasteriser against the CPU oracle, through the drop-in Python surface
(diff_gaussian_rasterization.GaussianRasterizer -> r2s_raster_forward C ABI).

Tolerance (BASELINE.json north_star): RGB / depth within 1e-4 relative.  The blend has hard
thresholds (alpha < 1/255, T(1-alpha) < 1e-4, the T = 0.5 median crossing, forward.cu:351-376), so a
1-ulp difference in exp() flips a discrete decision at a handful of pixels; those are counted and must
stay below 1e-4 of all pixels (SURVEY.md §8d "Parity gates")."""
import numpy as np
import pytest

from util_parity import record
from util_raster import compare_images, hip_render, oracle_render, scene_and_camera

pytestmark = pytest.mark.gpu

RTOL = 1e-4
MAX_BAD_FRAC = 1e-4


@pytest.mark.parametrize("P,W,H,cam,seed", [(3000, 320, 240, "side", 0), (20000, 640, 480, "side", 1),
                                            (20000, 640, 480, "wrist", 2), (6000, 331, 277, "side", 3)])
def test_forward_matches_oracle(P, W, H, cam, seed):
    sc, c = scene_and_camera(P, W, H, seed, cam=cam, bg=(0.1, 0.2, 0.3))
    n_ref, col_ref, radii_ref, dep_ref = oracle_render(sc, c)
    col, radii, dep = hip_render(sc, c)
    assert np.array_equal(radii, radii_ref)
    r = compare_images(col, dep, col_ref, dep_ref, rtol=RTOL)
    assert r["frac_rgb"] <= MAX_BAD_FRAC and r["frac_depth"] <= MAX_BAD_FRAC, r
    assert n_ref > 0


def test_num_rendered_and_intermediates_exact():
    import torch
    from r2s_hip.raster import RasterBatch

    sc, c = scene_and_camera(8000, 640, 480, 5)
    n_ref, _, radii_ref, _, dbg = oracle_render(sc, c, debug=True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    rb = RasterBatch(dev)
    s = rb.make_set(t(sc["means3D"]), t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    H, W = c["image_height"], c["image_width"]
    out_c = torch.empty(3, H, W, device=dev); out_d = torch.empty(1, H, W, device=dev)
    fr = dict(set=0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), campos=t(c["campos"]), bg=t(c["bg"]),
              tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], z_threshold=c["z_threshold"], out_color=out_c, out_depth=out_d)
    n, counts = rb.forward([s], [fr], W, H, want_counts=True)
    d = rb.debug()
    assert n == n_ref and counts == [n_ref]
    assert np.array_equal(d["radii"].numpy(), radii_ref)
    assert np.array_equal(d["tiles_touched"].numpy().astype(np.uint32), dbg["tiles_touched"])
    vis = radii_ref > 0
    np.testing.assert_allclose(d["depths"].numpy()[vis], dbg["depths"][vis], rtol=1e-6)
    np.testing.assert_allclose(d["geom"].numpy()[vis, 0:2], dbg["means2D"][vis], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(d["geom"].numpy()[vis][:, [2, 3, 4]], dbg["conic_opacity"][vis, :3], rtol=2e-4, atol=1e-6)
    # sorted instance list: identical wherever depths agree bit-for-bit (stable sort => unique order)
    same_depth = np.array_equal(d["depths"].numpy()[vis].view(np.uint32), dbg["depths"][vis].view(np.uint32))
    if same_depth:
        assert np.array_equal(d["point_list"].numpy().astype(np.uint32), dbg["point_list"])


def test_backward_auxiliaries_final_T_and_n_contrib_vs_oracle():
    """The compositor variant that also writes final_T / n_contrib (forward.cu:335, :380, :386-388; only on request): with the
    tile culling off the tile lists are the reference's, so n_contrib — the position of a pixel's last blended instance in its tile
    list — must equal the oracle's exactly, final_T to float tolerance, and colour / depth must be the bits of the plain variant."""
    import torch
    from r2s_hip.raster import RasterBatch

    sc, c = scene_and_camera(6000, 320, 240, 11)
    _, col_ref, _, dep_ref, dbg = oracle_render(sc, c, debug=True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    H, W = c["image_height"], c["image_width"]
    outs = []
    for aux in (False, True):
        rb = RasterBatch(dev)
        rb.set_tile_culling(False)
        s = rb.make_set(t(sc["means3D"]), t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
        out_c = torch.empty(3, H, W, device=dev); out_d = torch.empty(1, H, W, device=dev)
        fT = torch.full((1, H, W), -1.0, device=dev); nC = torch.full((1, H, W), -1, dtype=torch.int32, device=dev)
        if aux:
            rb.set_aux(fT, nC)
        fr = dict(set=0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), campos=t(c["campos"]), bg=t(c["bg"]),
                  tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], z_threshold=c["z_threshold"], out_color=out_c, out_depth=out_d)
        rb.forward([s], [fr], W, H)
        torch.cuda.synchronize()
        outs.append((out_c.cpu().numpy(), out_d.cpu().numpy(), fT.cpu().numpy()[0], nC.cpu().numpy()[0]))
    (c0, d0, _, _), (c1, d1, fT, nC) = outs
    assert np.array_equal(c0, c1) and np.array_equal(d0, d1), "the auxiliary outputs must not change a pixel"
    n_ref, T_ref = np.asarray(dbg["n_contrib"]).astype(np.int64), np.asarray(dbg["final_T"])
    agree = float((nC == n_ref).mean())
    record("final_T / n_contrib vs oracle", n_contrib_agree_frac=agree, final_T_max_abs=float(np.abs(fT - T_ref)[nC == n_ref].max()), tol=1e-4)
    assert agree >= 1.0 - 5e-4, agree      # a threshold decision (alpha < 1/255, T < 1e-4) can flip on a float-rounded alpha at a few pixels
    assert np.abs(fT - T_ref)[nC == n_ref].max() < 1e-4
    assert (nC >= 0).all() and (fT >= 0).all()


def test_p_zero_returns_zero_images():
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    from util_raster import torch_settings
    from r2s_hip import synth

    c = synth.side_camera(64, 48, bg=(0.5, 0.5, 0.5))
    cam = torch_settings(c, "cuda:0")
    e = torch.zeros(0, 3, device="cuda:0")
    im, radii, depth = GaussianRasterizer(cam)(means3D=e, means2D=e, opacities=torch.zeros(0, 1, device="cuda:0"),
                                               shs=torch.zeros(0, 1, 3, device="cuda:0"), scales=e, rotations=torch.zeros(0, 4, device="cuda:0"))
    # rasterize_points.cu:68-70,82: P == 0 -> zero-initialised outputs, NOT background
    assert float(im.abs().max()) == 0.0 and float(depth.abs().max()) == 0.0 and radii.numel() == 0


def test_all_culled_writes_background_and_default_depth():
    sc, c = scene_and_camera(500, 64, 48, 9, bg=(0.25, 0.5, 0.75))
    sc["means3D"] = sc["means3D"].copy()
    sc["means3D"][:, 0] += 100.0  # behind the side camera -> p_view.z <= z_threshold
    col, radii, dep = hip_render(sc, c)
    assert (radii == 0).all()
    assert np.allclose(col[0], 0.25) and np.allclose(col[1], 0.5) and np.allclose(col[2], 0.75)
    assert (dep == 15.0).all()


def test_argument_checks_match_reference():
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    from util_raster import torch_settings
    from r2s_hip import synth

    cam = torch_settings(synth.side_camera(64, 48), "cuda:0")
    m = torch.zeros(4, 3, device="cuda:0")
    r = GaussianRasterizer(cam)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=m[:, :1])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=m[:, :1], shs=torch.zeros(4, 1, 3, device="cuda:0"))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        r(means3D=torch.zeros(4, 2, device="cuda:0"), means2D=m, opacities=m[:, :1], shs=torch.zeros(4, 1, 3, device="cuda:0"),
          scales=m, rotations=torch.zeros(4, 4, device="cuda:0"))


def test_sh_degree3_and_precomputed_paths():
    sc, c = scene_and_camera(4000, 320, 240, 11, sh_coeffs=16, sh_degree=3)
    _, col_ref, radii_ref, dep_ref = oracle_render(sc, c)
    col, radii, dep = hip_render(sc, c)
    assert np.array_equal(radii, radii_ref)
    r = compare_images(col, dep, col_ref, dep_ref)
    assert r["frac_rgb"] <= 2e-4, r
    # colors_precomp + cov3D_precomp
    rng = np.random.default_rng(0)
    sc2 = dict(means3D=sc["means3D"], opacities=sc["opacities"], colors_precomp=rng.uniform(0, 1, (4000, 3)).astype(np.float32))
    A = rng.normal(0, 0.004, (4000, 3, 3)).astype(np.float32)
    S = A @ A.transpose(0, 2, 1)
    sc2["cov3D_precomp"] = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    _, col_ref, radii_ref, dep_ref = oracle_render(sc2, c)
    col, radii, dep = hip_render(sc2, c)
    assert np.array_equal(radii, radii_ref)
    r = compare_images(col, dep, col_ref, dep_ref)
    assert r["frac_rgb"] <= 2e-4, r


def test_prefiltered_trap_is_reported():
    from r2s_hip import R2SError

    sc, c = scene_and_camera(200, 64, 48, 4)
    sc["means3D"] = sc["means3D"].copy(); sc["means3D"][0, 0] += 100.0
    c = dict(c); c["prefiltered"] = True
    with pytest.raises(R2SError, match="prefiltered"):
        hip_render(sc, c)


def _batch_render(sc, cams, cull, device="cuda:0"):
    import torch
    from r2s_hip.raster import RasterBatch

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    rb = RasterBatch(device)
    rb.set_tile_culling(cull)
    s = rb.make_set(t(sc["means3D"]), t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    H, W = cams[0]["image_height"], cams[0]["image_width"]
    frames, keep = [], []
    for c in cams:
        oc = torch.empty(3, H, W, device=device); od = torch.empty(1, H, W, device=device)
        keep.append((oc, od))
        frames.append(dict(set=0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), campos=t(c["campos"]), bg=t(c["bg"]),
                           tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], z_threshold=c["z_threshold"], out_color=oc, out_depth=od))
    n, counts = rb.forward([s], frames, W, H, want_counts=True)
    torch.cuda.synchronize()
    return n, counts, [(a.cpu().numpy(), b.cpu().numpy()) for a, b in keep]


def test_batched_frames_match_single_frame_calls_and_tile_culling_is_output_exact():
    from r2s_hip import synth

    sc, _ = scene_and_camera(12000, 640, 480, 21)
    cams = [synth.side_camera(640, 480, bg=(0.1, 0.2, 0.3)), synth.wrist_camera(640, 480, bg=(0.3, 0.2, 0.1))]
    n0, counts0, img0 = _batch_render(sc, cams, cull=False)
    n1, counts1, img1 = _batch_render(sc, cams, cull=True)
    singles = [hip_render(sc, c) for c in cams]
    for (c0, d0), (c1, d1), (cs, _, ds) in zip(img0, img1, singles):
        assert np.array_equal(c0, cs) and np.array_equal(d0, ds)      # batch == single-frame drop-in, bit for bit
        assert np.array_equal(c0, c1) and np.array_equal(d0, d1)      # culling never changes a pixel
    assert sum(counts0) == n0 and sum(counts1) == n1
    assert n1 < n0, (n0, n1)
    # and the oracle agrees with both
    for cam, (c1, d1) in zip(cams, img1):
        _, col_ref, _, dep_ref = oracle_render(sc, cam)
        r = compare_images(c1, d1, col_ref, dep_ref)
        assert r["frac_rgb"] <= MAX_BAD_FRAC and r["frac_depth"] <= MAX_BAD_FRAC, r


def test_sync_free_batches_are_bit_identical_and_report_overflow():
    """r2s_raster_ctx_set_async: after the first batch nothing is read back between scan and emit (the reference blocks on a
    cudaMemcpy there, rasterizer_impl.cu:284).  Images must equal the synchronous mode bit for bit; a batch that outgrows the
    capacity derived from its predecessor is reported, and the batch after it is correct again."""
    import torch
    from r2s_hip import synth
    from r2s_hip.raster import RasterBatch

    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    W, H = 320, 240
    cams = [synth.side_camera(W, H), synth.wrist_camera(W, H)]

    def batch(rb, scene, out_c, out_d):
        s = rb.make_set(t(scene["means3D"]), t(scene["opacities"]), shs=t(scene["shs"]), scales=t(scene["scales"]), rotations=t(scene["rotations"]))
        frames = [dict(set=0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), campos=t(c["campos"]), bg=t(c["bg"]), tanfovx=c["tanfovx"],
                       tanfovy=c["tanfovy"], z_threshold=c["z_threshold"], out_color=out_c[v], out_depth=out_d[v]) for v, c in enumerate(cams)]
        return rb.prepare([s], frames)

    small, big = synth.gaussian_scene(3000, 90), synth.gaussian_scene(30000, 91)
    ref = {}
    for name, sc in (("small", small), ("big", big)):
        rb = RasterBatch(dev)
        oc = torch.zeros(2, 3, H, W, device=dev); od = torch.zeros(2, 1, H, W, device=dev)
        n = rb.forward(batch(rb, sc, oc, od), None, W, H)
        torch.cuda.synchronize()
        ref[name] = (n, oc.clone(), od.clone())
    rb = RasterBatch(dev)
    rb.set_async(True)
    oc = torch.zeros(2, 3, H, W, device=dev); od = torch.zeros(2, 1, H, W, device=dev)
    prep_small, prep_big = batch(rb, small, oc, od), batch(rb, big, oc, od)
    for k in range(3):          # first call synchronises once, the next two are sync-free
        oc.zero_(); od.zero_()
        n = rb.forward(prep_small, None, W, H)
        torch.cuda.synchronize()
        assert n == ref["small"][0]
        assert torch.equal(oc, ref["small"][1]) and torch.equal(od, ref["small"][2]), k
    running, n_seen, overflows = rb.poll(wait=True)
    assert not running and n_seen == ref["small"][0] and overflows == 0
    # ten times more instances than the capacity was sized for: flagged, never written out of bounds
    rb.forward(prep_big, None, W, H)
    running, n_seen, overflows = rb.poll(wait=True)
    assert overflows == 1 and n_seen == ref["big"][0]
    oc.zero_(); od.zero_()
    n = rb.forward(prep_big, None, W, H)   # re-sizes (one synchronisation) and renders correctly
    torch.cuda.synchronize()
    assert n == ref["big"][0] and torch.equal(oc, ref["big"][1]) and torch.equal(od, ref["big"][2])
    oc.zero_(); od.zero_()
    rb.forward(prep_big, None, W, H)       # sync-free again
    torch.cuda.synchronize()
    assert torch.equal(oc, ref["big"][1]) and torch.equal(od, ref["big"][2]) and rb.poll(wait=True)[2] == 1


def test_inline_assembly_blend_equals_its_cxx_specification_bit_for_bit():
    """k_composite's blend is inline assembly (predicates in EXEC, raster.hip); its C++ form stays in
    the source as the specification (-DR2S_COMP_CXX).  Both builds ship (csrc/Makefile: libr2s_hip.so, libr2s_hip_cxx.so); the same
    frames rendered through each — single-frame and batched entry points, full and ragged tiles, the reference's 848x480 frame —
    must be bit-identical, so that a compiler bump that changes a register constraint cannot silently change an image."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "real2sim-eval_amd")
    libs = {"asm": os.path.join(pkg, "libr2s_hip.so"), "cxx": os.path.join(pkg, "libr2s_hip_cxx.so")}
    assert os.path.exists(libs["cxx"]), "libr2s_hip_cxx.so missing: __graft_entry__.build() builds it next to libr2s_hip.so"
    got = {}
    for k, path in libs.items():
        out = subprocess.run([sys.executable, os.path.join(here, "frame_hash_helper.py")], env=dict(os.environ, R2S_HIP_LIB=path), capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        got[k] = {l.split()[1]: l.split()[2] for l in out.stdout.splitlines() if l.startswith("HASH ")}
    assert set(got["asm"]) == {"side", "wrist", "ref_frame", "ragged", "rollout"} == set(got["cxx"])
    assert got["asm"] == got["cxx"], (got["asm"], got["cxx"])
