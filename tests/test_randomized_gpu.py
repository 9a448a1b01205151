"""Seed-swept randomized parity (HIP vs oracle): random cameras / splat statistics for the rasteriser, random material
parameters / initial velocities for the stepper.  Small sizes so the oracle stays fast."""
import numpy as np
import pytest

from util_physics import hip_env, make_object, oracle_env
from util_raster import compare_images, hip_render, oracle_render
from util_parity import close

pytestmark = pytest.mark.gpu


def _look_at(eye, target, up=(0, 0, 1)):
    eye, target, up = map(lambda a: np.asarray(a, np.float64), (eye, target, up))
    f = target - eye; f /= np.linalg.norm(f)
    r = np.cross(f, up); r /= np.linalg.norm(r)
    d = np.cross(f, r)  # camera y axis points down the image
    c2w = np.eye(4); c2w[:3, 0] = r; c2w[:3, 1] = d; c2w[:3, 2] = f; c2w[:3, 3] = eye
    return np.linalg.inv(c2w)


@pytest.mark.parametrize("seed", range(6))
def test_random_camera_and_splat_statistics(seed):
    from r2s_hip import synth

    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(97, 400)), int(rng.integers(64, 300))
    P = int(rng.integers(500, 6000))
    f = float(rng.uniform(0.6, 1.6) * W)
    K = [[f, 0, W / 2 + rng.uniform(-10, 10)], [0, f * rng.uniform(0.9, 1.1), H / 2 + rng.uniform(-10, 10)], [0, 0, 1]]
    eye = np.array([rng.uniform(0.5, 1.2), rng.uniform(-0.5, 0.5), rng.uniform(0.2, 0.8)])
    cam = synth.camera_settings(K, _look_at(eye, (0.37, 0.05, 0.0)), W, H, z_threshold=float(rng.uniform(0.01, 0.3)),
                                bg=tuple(rng.uniform(0, 1, 3)), sh_degree=int(rng.integers(0, 4)))
    sc = synth.gaussian_scene(P, seed, sh_coeffs=16)
    sc["scales"] = (sc["scales"] * rng.uniform(0.3, 4.0)).astype(np.float32)      # from sub-pixel to many-tile splats
    sc["opacities"] = np.clip(sc["opacities"] * rng.uniform(0.3, 1.2), 0, 1).astype(np.float32)
    _, col_ref, radii_ref, dep_ref = oracle_render(sc, cam)
    col, radii, dep = hip_render(sc, cam)
    assert np.array_equal(radii, radii_ref)
    r = compare_images(col, dep, col_ref, dep_ref)
    assert r["frac_rgb"] <= 3e-4 and r["frac_depth"] <= 3e-4, r


@pytest.mark.parametrize("seed", range(5))
def test_random_material_parameters_and_velocities(seed):
    rng = np.random.default_rng(2000 + seed)
    shape = ["rope", "sloth", "T", "cloth"][seed % 4]
    ob = make_object(shape, int(rng.integers(300, 1200)), seed=seed, lift=float(rng.uniform(0.0, 0.01)))
    ob["v0"] = rng.normal(0, 0.3, ob["points"].shape).astype(np.float32)
    kw = dict(dashpot_damping=float(rng.uniform(10, 200)), drag_damping=float(rng.uniform(0, 10)),
              collide_elas=float(rng.uniform(-0.2, 1.3)), collide_fric=float(rng.uniform(-0.5, 2.5)),
              spring_Y_max=float(rng.choice([1e5, 2e4])), collision_dist=float(rng.choice([0.005, 0.008])),
              collide_self_elas=float(rng.uniform(0, 1)), collide_self_fric=float(rng.uniform(0, 2)))
    n_sub = 80
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    for _ in range(2):
        o.update_collision_graph(); h.update_collision_graph()
        o.step(); h.step()
    assert close(h.x[0].cpu().numpy(), o.x, 1e-5)
    assert close(h.v[0].cpu().numpy(), o.v, 5e-3)
