"""Known-answer tests that pin the CPU raster oracle to the reference's formulas.
The reference ships no tests or golden vectors, so each case below is a closed form derived from the cited
lines of third-party/diff-gaussian-rasterization-w-depth/cuda_rasterizer/{forward.cu,auxiliary.h,rasterizer_impl.cu}."""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
C0 = 0.28209479177387814


def pinhole(W=64, H=48, f=50.0, z_threshold=0.05, bg=(0.1, 0.2, 0.3)):
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    return oracle.setup_camera(W, H, K, np.eye(4), z_threshold=z_threshold, bg=bg)


def render(cam, means, opac, sh0, scales, rots=None, **kw):
    means = np.asarray(means, np.float32).reshape(-1, 3)
    P = len(means)
    rots = np.tile([1, 0, 0, 0], (P, 1)) if rots is None else rots
    return oracle.raster_forward(means, np.asarray(opac, np.float32).reshape(P, 1), cam["viewmatrix"], cam["projmatrix"],
                                 cam["campos"], cam["tanfovx"], cam["tanfovy"], cam["image_height"], cam["image_width"],
                                 cam["bg"], shs=np.asarray(sh0, np.float32).reshape(P, 1, 3), scales=np.asarray(scales, np.float32).reshape(P, 3),
                                 rotations=np.asarray(rots, np.float32), z_threshold=cam["z_threshold"], debug=True, **kw)


def test_K1_single_isotropic_gaussian_on_axis():
    cam = pinhole()
    z, sig, o, f = 2.0, 0.05, 0.8, 50.0
    sh = np.array([1.0, 0.0, -3.0])
    n, col, radii, dep, d = render(cam, [[0, 0, z]], [o], [sh], [[sig] * 3])
    var = sig**2 * f**2 / z**2 + 0.3  # forward.cu:106-111
    assert np.allclose(d["conic_opacity"][0], [1 / var, 0, 1 / var, o], rtol=1e-5, atol=1e-7)
    assert radii[0] == int(np.ceil(3 * np.sqrt(var)))  # forward.cu:230-233 (lambda1 = var + sqrt(0.1) floor not hit? see below)
    # ndc2Pix(0, S) = (S - 1) / 2, auxiliary.h:41-44
    assert np.allclose(d["means2D"][0], [31.5, 23.5])
    rgb = np.maximum(0.0, C0 * sh + 0.5)  # forward.cu:30,63,70
    assert np.allclose(d["rgb"][0], rgb, rtol=1e-6)
    # nearest pixels (31,23),(32,24): d = (+-0.5, +-0.5)
    power = -0.5 * (0.25 / var + 0.25 / var)
    alpha = min(0.99, o * np.exp(power))
    exp_col = rgb * alpha + (1 - alpha) * cam["bg"]
    assert np.allclose(col[:, 23, 31], exp_col, rtol=1e-5)
    assert np.allclose(col[:, 24, 32], exp_col, rtol=1e-5)
    assert dep[0, 23, 31] == pytest.approx(z) and alpha > 0.5  # median depth set when T crosses 0.5 (forward.cu:369-373)
    # far pixel: untouched tile region -> bg and default depth 15 (forward.cu:309)
    assert np.allclose(col[:, 0, 0], cam["bg"]) and dep[0, 0, 0] == 15.0
    assert d["n_contrib"][23, 31] == 1 and d["final_T"][23, 31] == pytest.approx(1 - alpha, rel=1e-5)


def test_K1b_radius_uses_eigenvalue_floor():
    # mid^2 - det = 0 for an isotropic splat -> sqrt(max(0.1, 0)) = 0.316: lambda1 = var + 0.316 (forward.cu:231)
    cam = pinhole()
    n, col, radii, dep, d = render(cam, [[0, 0, 2.0]], [0.5], [[0, 0, 0]], [[0.05] * 3])
    var = 0.05**2 * 50.0**2 / 4.0 + 0.3
    assert radii[0] == int(np.ceil(3 * np.sqrt(var + np.sqrt(0.1))))


def test_K1c_low_opacity_leaves_default_depth():
    cam = pinhole()
    n, col, radii, dep, d = render(cam, [[0, 0, 2.0]], [0.4], [[1, 1, 1]], [[0.05] * 3])
    assert (dep == 15.0).all()  # alpha <= 0.4 never takes T below 0.5


def test_K2_input_order_does_not_matter():
    cam = pinhole()
    rng = np.random.default_rng(0)
    means = np.stack([rng.uniform(-0.3, 0.3, 40), rng.uniform(-0.2, 0.2, 40), rng.uniform(1.0, 3.0, 40)], 1)
    op, sh, sc = rng.uniform(0.2, 0.95, 40), rng.normal(0, 1, (40, 3)), rng.uniform(0.02, 0.08, (40, 3))
    a = render(cam, means, op, sh, sc)
    perm = rng.permutation(40)
    b = render(cam, means[perm], op[perm], sh[perm], sc[perm])
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])


def test_K2b_equal_depth_ties_resolve_by_index():
    cam = pinhole()
    # two coincident-depth splats with different colours: stable sort keeps index order (rasterizer_impl.cu:306-311)
    n, col, radii, dep, d = render(cam, [[0, 0, 2.0], [0, 0, 2.0]], [0.6, 0.6], [[3, -2, -2], [-2, 3, -2]], [[0.05] * 3] * 2)
    assert list(d["point_list"][:2]) == [0, 1]
    a0 = 0.6 * np.exp(-0.5 * 0.5 / (0.05**2 * 2500 / 4 + 0.3))
    rgb0 = np.maximum(0, C0 * np.array([3, -2, -2.0]) + 0.5); rgb1 = np.maximum(0, C0 * np.array([-2, 3, -2.0]) + 0.5)
    exp = rgb0 * a0 + rgb1 * a0 * (1 - a0) + (1 - a0) ** 2 * cam["bg"]
    assert np.allclose(col[:, 23, 31], exp, rtol=1e-5)


def test_K3_z_threshold_is_inclusive_cull():
    cam = pinhole(z_threshold=0.5)
    n, _, radii, _, _ = render(cam, [[0, 0, 0.5], [0, 0, np.nextafter(np.float32(0.5), np.float32(1))]], [0.5, 0.5], [[0, 0, 0]] * 2, [[0.01] * 3] * 2)
    assert radii[0] == 0 and radii[1] > 0  # `p_view.z <= z_threshold` culls, auxiliary.h:155


def test_K4_tiles_touched_is_rect_area():
    cam = pinhole(W=96, H=64)
    # centre at pixel (47.5, 31.5); radius r -> rect per getRect (auxiliary.h:46-56)
    n, _, radii, _, d = render(cam, [[0, 0, 2.0]], [0.5], [[0, 0, 0]], [[0.2] * 3])
    r = radii[0]
    px, py = d["means2D"][0]
    x0, x1 = max(0, int((px - r) / 16)), min(6, int((px + r + 15) / 16))
    y0, y1 = max(0, int((py - r) / 16)), min(4, int((py + r + 15) / 16))
    assert d["tiles_touched"][0] == (x1 - x0) * (y1 - y0) == n
    # off-screen splat whose rect is empty is skipped after the radius is computed (forward.cu:237-238)
    n2, _, radii2, _, _ = render(cam, [[50.0, 0, 2.0]], [0.5], [[0, 0, 0]], [[0.01] * 3])
    assert n2 == 0 and radii2[0] == 0


def test_K5_opaque_stack_terminates_before_blending():
    cam = pinhole()
    P = 12
    means = [[0, 0, 1.0 + 0.1 * k] for k in range(P)]
    n, col, _, dep, d = render(cam, means, [0.99] * P, [[1, 1, 1]] * P, [[0.5] * 3] * P)
    # alpha ~ 0.99 near the centre: T ~ 0.01^k; T(1-alpha) < 1e-4 first at the third Gaussian -> only 2 are
    # blended and the terminating one is NOT (forward.cu:353-358)
    T, blended = 1.0, 0
    for k in range(P):
        var = 0.5**2 * 50.0**2 / (1.0 + 0.1 * k) ** 2 + 0.3
        alpha = min(0.99, 0.99 * np.exp(-0.5 * (0.25 + 0.25) / var))
        if T * (1 - alpha) < 1e-4:
            break
        T *= 1 - alpha
        blended += 1
    assert blended == 2 and d["n_contrib"][24, 32] == 2
    assert d["final_T"][24, 32] == pytest.approx(T, rel=1e-4)
    assert dep[0, 24, 32] == pytest.approx(1.0)  # first Gaussian takes T from 1 to ~0.01 (crosses 0.5)


def test_K6_alpha_and_power_skips():
    cam = pinhole()
    # opacity below 1/255 never contributes (forward.cu:351)
    n, col, _, dep, d = render(cam, [[0, 0, 2.0]], [1.0 / 256.0], [[5, 5, 5]], [[0.05] * 3])
    assert np.allclose(col, np.asarray(cam["bg"])[:, None, None]) and (d["n_contrib"] == 0).all()


def test_K7_sh_dc_negative_clamp():
    cam = pinhole()
    _, _, _, _, d = render(cam, [[0, 0, 2.0]], [0.5], [[-5.0, 0.0, 5.0]], [[0.05] * 3])
    assert np.allclose(d["rgb"][0], [0.0, 0.5, C0 * 5 + 0.5], rtol=1e-6)


def test_K8_zero_gaussians_gives_zero_not_background():
    cam = pinhole()
    n, col, radii, dep = oracle.raster_forward(np.zeros((0, 3)), np.zeros((0, 1)), cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                                              cam["tanfovx"], cam["tanfovy"], 48, 64, cam["bg"], shs=np.zeros((0, 1, 3)),
                                              scales=np.zeros((0, 3)), rotations=np.zeros((0, 4)))
    assert n == 0 and (col == 0).all() and (dep == 0).all()  # rasterize_points.cu:68-70,82


def test_K9_camera_from_reference_config_golden():
    """setup_camera (sim/utils/gs/transform_utils.py:7-31) on the side camera of cfg/env/xarm_gripper.yaml:25-35.
    The golden values were produced by tests/golden/make_camera_golden.py (torch float32 ops, same formulas)."""
    from r2s_hip import synth

    g = json.load(open(os.path.join(HERE, "golden", "camera_side_848x480.json")))
    cam = oracle.setup_camera(848, 480, synth.SIDE_K, np.linalg.inv(synth.SIDE_C2W), z_threshold=0.05)
    assert cam["tanfovx"] == pytest.approx(g["tanfovx"], rel=1e-7) and cam["tanfovy"] == pytest.approx(g["tanfovy"], rel=1e-7)
    assert np.allclose(cam["viewmatrix"].reshape(-1), g["viewmatrix"], rtol=1e-6, atol=1e-7)
    assert np.allclose(cam["projmatrix"].reshape(-1), g["projmatrix"], rtol=1e-5, atol=1e-6)
    assert np.allclose(cam["campos"], g["campos"], rtol=1e-5, atol=1e-6)
    # closed forms: tanfov = w / (2 fx); view = w2c^T; proj[2][3] == 1 column structure
    assert cam["tanfovx"] == pytest.approx(848 / (2 * 427.2920227050781))
    assert np.allclose(cam["viewmatrix"][0].T, np.linalg.inv(synth.SIDE_C2W), atol=1e-6)
    # synth.camera_settings is the same restatement used by bench/tests on the HIP side
    c2 = synth.side_camera(848, 480)
    assert np.array_equal(c2["viewmatrix"], cam["viewmatrix"]) and np.array_equal(c2["projmatrix"], cam["projmatrix"])


def test_higher_msb_sort_bits():
    # getHigherMsb, rasterizer_impl.cu:35-50: tiles 300 -> 9, 1200 -> 11, 1590 -> 11, 3600 -> 12
    assert [oracle.higher_msb(n) for n in (300, 1200, 1590, 3600)] == [9, 11, 11, 12]


def test_quaternion_is_not_renormalised():
    cam = pinhole()
    a = render(cam, [[0, 0, 2.0]], [0.5], [[0, 0, 0]], [[0.05, 0.02, 0.03]], rots=np.array([[1, 0, 0, 0.0]]))
    b = render(cam, [[0, 0, 2.0]], [0.5], [[0, 0, 0]], [[0.05, 0.02, 0.03]], rots=np.array([[2, 0, 0, 0.0]]))
    # R(2,0,0,0) = I as well (only r^2-free terms), so identical; a non-unit (1,1,0,0) is NOT a rotation:
    c = render(cam, [[0, 0, 2.0]], [0.5], [[0, 0, 0]], [[0.05, 0.02, 0.03]], rots=np.array([[1, 1, 0, 0.0]]))
    assert np.array_equal(a[4]["conic_opacity"], b[4]["conic_opacity"])
    # forward.cu:127: R = [[1,0,0],[0,-1,-2],[0,2,-1]] for q=(1,1,0,0): cov grows by |q|^4 on the y/z block
    assert not np.allclose(a[4]["conic_opacity"], c[4]["conic_opacity"])


def test_f64_shadow_agrees_with_f32_oracle():
    from util_raster import compare_images, oracle_render, scene_and_camera

    sc, c = scene_and_camera(1500, 160, 120, 0)
    _, c32, r32, d32 = oracle_render(sc, c)
    _, c64, r64, d64 = oracle_render(sc, c, f64=True)
    r = compare_images(c32, d32, c64, d64)
    assert r["frac_rgb"] < 2e-3 and (r32 != r64).mean() < 1e-3, r


def test_prefiltered_trap():
    cam = pinhole()
    with pytest.raises(RuntimeError, match="prefiltered"):
        render(cam, [[0, 0, -1.0]], [0.5], [[0, 0, 0]], [[0.05] * 3], prefiltered=True)
