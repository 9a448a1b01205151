"""Full-size (BASELINE.json configs) checks on the GPU.

Where the CPU oracle still finishes in seconds the HIP path is compared with it directly at full size; beyond that,
size-independent properties are asserted: determinism, batch-index independence, permutation invariance of the
render, batched == single-frame calls, free-space momentum balance of the spring network."""
import numpy as np
import pytest

from util_physics import hip_env, make_object, oracle_env
from util_raster import assert_image_gate, compare_images, hip_render, oracle_render, scene_and_camera
from util_parity import close

pytestmark = pytest.mark.gpu


def test_C1_rope_scene_full_resolution_vs_oracle():
    """configs[1]: ~40k Gaussians, 640x480, both cameras."""
    for cam in ("side", "wrist"):
        sc, c = scene_and_camera(40000, 640, 480, 31, cam=cam)
        n_ref, col_ref, radii_ref, dep_ref, frag = oracle_render(sc, c, fragile=True)
        col, radii, dep = hip_render(sc, c)
        assert np.array_equal(radii, radii_ref)
        r = compare_images(col, dep, col_ref, dep_ref, fragile=frag, what=f"C1 40k Gaussians 640x480 {cam} camera vs oracle")
        assert_image_gate(r, 640 * 480, cam)


def test_C2_sloth_scene_80k_gaussians_vs_oracle_and_permutation_invariance():
    sc, c = scene_and_camera(80000, 640, 480, 32)
    _, col_ref, radii_ref, dep_ref, frag = oracle_render(sc, c, fragile=True)
    col, radii, dep = hip_render(sc, c)
    assert np.array_equal(radii, radii_ref)
    r = compare_images(col, dep, col_ref, dep_ref, fragile=frag, what="C2 80k Gaussians 640x480 vs oracle")
    assert_image_gate(r, 640 * 480)
    # the image does not depend on the order Gaussians are given in, except where two splats share the exact float
    # depth (the stable sort then keeps index order, rasterizer_impl.cu:306-311; with 80k splats a few hundred pairs
    # collide in float32, so compare to rounding, not bitwise)
    perm = np.random.default_rng(0).permutation(80000)
    sc2 = {k: v[perm] for k, v in sc.items()}
    col2, radii2, dep2 = hip_render(sc2, c)
    assert np.array_equal(radii2, radii[perm])
    r = compare_images(col2, dep2, col, dep, rtol=1e-5, atol=1e-5)
    assert r["frac_rgb"] <= 1e-4 and r["frac_depth"] <= 1e-4, r
    # and it is deterministic run to run
    col3, _, dep3 = hip_render(sc, c)
    assert np.array_equal(col3, col) and np.array_equal(dep3, dep)


def test_C4_multi_view_1280x720_batched_equals_single_calls():
    import torch
    from r2s_hip import synth
    from r2s_hip.raster import RasterBatch

    dev = "cuda:0"
    W, H = 1280, 720
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    cams = [synth.side_camera(W, H), synth.wrist_camera(W, H), synth.wrist_camera(W, H, eef_pos=(0.45, 0.0, 0.5)),
            synth.wrist_camera(W, H, eef_pos=(0.30, 0.10, 0.45))]
    scenes = [synth.gaussian_scene(30000, 40 + e) for e in range(2)]
    rb = RasterBatch(dev)
    sets = [rb.make_set(t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]), rotations=t(s["rotations"])) for s in scenes]
    out_c = torch.empty(2, 4, 3, H, W, device=dev); out_d = torch.empty(2, 4, 1, H, W, device=dev)
    frames = []
    for e in range(2):
        for v, c in enumerate(cams):
            frames.append(dict(set=e, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), campos=t(c["campos"]), bg=t(c["bg"]),
                               tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], z_threshold=c["z_threshold"], out_color=out_c[e, v], out_depth=out_d[e, v]))
    n = rb.forward(sets, frames, W, H)
    torch.cuda.synchronize()
    assert n > 0
    for e in range(2):
        for v, c in enumerate(cams):
            col, _, dep = hip_render(scenes[e], c)
            assert np.array_equal(out_c[e, v].cpu().numpy(), col) and np.array_equal(out_d[e, v].cpu().numpy(), dep), (e, v)
    # one view against the oracle at 1280x720
    _, col_ref, _, dep_ref, frag = oracle_render(scenes[1], cams[0], fragile=True)
    r = compare_images(out_c[1, 0].cpu().numpy(), out_d[1, 0].cpu().numpy(), col_ref, dep_ref, fragile=frag, what="C4 one view 1280x720 vs oracle")
    assert_image_gate(r, W * H)


def test_C2_sloth_15k_particles_one_env_vs_oracle_short_horizon():
    ob = make_object("sloth", 15000, seed=50)
    o = oracle_env(ob, num_substeps=40, self_collision=False)
    h = hip_env(ob, num_substeps=40, self_collision=False)
    o.step(); h.step()
    assert close(h.x[0].cpu().numpy(), o.x, 1e-5)
    assert np.abs(o.x - ob["points"]).max() > 1e-5


def test_C3_T_block_real_size_vs_oracle_with_self_collision():
    ob = make_object("T", 2229, seed=51, lift=0.002)
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = -0.3
    o = oracle_env(ob, num_substeps=150)
    h = hip_env(ob, num_substeps=150)
    for _ in range(2):
        o.update_collision_graph(); h.update_collision_graph()
        o.step(); h.step()
    assert close(h.x[0].cpu().numpy(), o.x, 1e-5)


def test_batched_32_envs_bitwise_independent_of_env_index_and_deterministic():
    import torch

    ob = make_object("sloth", 15000, seed=52, lift=0.05)
    h = hip_env(ob, num_substeps=60, n_env=32)
    h.update_collision_graph(); h.step()
    x = h.x.cpu().numpy()
    for e in range(1, 32):
        assert np.array_equal(x[e], x[0]), e          # identical inputs -> identical bits, whichever slot
    h2 = hip_env(ob, num_substeps=60, n_env=32)
    h2.update_collision_graph(); h2.step()
    assert np.array_equal(h2.x.cpu().numpy(), x)      # and run to run
    # a single-environment handle gives the same trajectory as slot 0 of the batch
    h1 = hip_env(ob, num_substeps=60, n_env=1)
    h1.update_collision_graph(); h1.step()
    assert np.array_equal(h1.x[0].cpu().numpy(), x[0])


def test_spring_forces_conserve_momentum_in_free_space():
    """Internal spring + dashpot forces are pairwise antisymmetric: with gravity and drag the centre-of-mass velocity
    follows v <- (v + g dt) exp(-drag dt) exactly, whatever the springs do."""
    ob = make_object("sloth", 15000, seed=53, lift=0.5)
    rng = np.random.default_rng(1)
    ob["points"] = ob["points"] + rng.normal(0, 2e-4, ob["points"].shape).astype(np.float32)  # strain the network
    n = 200
    h = hip_env(ob, num_substeps=n, self_collision=False)
    h.step()
    dt, drag, g = 5e-5, 3.0, -9.8
    v = 0.0
    for _ in range(n):
        v = (v + g * dt) * np.exp(-dt * drag)
    vc = h.v[0].double().mean(0).cpu().numpy()
    assert abs(vc[2] - v) < 2e-5 and abs(vc[0]) < 2e-5 and abs(vc[1]) < 2e-5


def test_reference_default_frame_848x480_side_and_wrist_cameras_vs_oracle():
    """VERDICT r3 item 7 / missing #5: the reference renders 848x480 (53x30 tiles: a ragged last tile column) from the fixed side
    camera and the wrist camera on the gripper (cfg/env/xarm_gripper.yaml:21-49, env.py:55-56).  Two environments of the headline
    scene at that frame size through the batched rollout (skinning -> per-environment wrist camera -> raster), after two env steps;
    every frame of environment 1 against the raster oracle under the classified image gate."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    W, H = 848, 480
    ro = BatchedRollout("sloth_32env", n_env=2, num_substeps=30, seed=9, res=(W, H), close_at=1, settle_steps=0)
    assert (ro.W, ro.H) == (W, H)
    for _ in range(2):
        ro.step()
    col, dep = ro.observations()
    torch.cuda.synchronize()
    assert ro.lossy_batches == 0
    sc = ro.scene_numpy(1)
    for v, name in ((0, "side"), (1, "wrist")):
        c = ro.camera_numpy(1, v)
        assert c["image_width"] == W and c["image_height"] == H
        n_ref, col_ref, _, dep_ref, frag = oracle_render(sc, c, fragile=True)
        assert n_ref > 0 and col_ref.std() > 0
        r = compare_images(col[1, v].cpu().numpy(), dep[1, v].cpu().numpy(), col_ref, dep_ref, fragile=frag, what=f"848x480 {name} camera, env 1, vs oracle")
        assert_image_gate(r, W * H, name)
    # the side camera at this size IS the calibrated camera of the reference (no rescaling): its matrices are the pinned fixture's
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "camera_side_848x480.json")))
    c0 = ro.camera_numpy(0, 0)
    assert np.allclose(np.asarray(fx["viewmatrix"], np.float32).reshape(4, 4), c0["viewmatrix"].reshape(4, 4), atol=1e-6)
    assert np.allclose(np.asarray(fx["projmatrix"], np.float32).reshape(4, 4), c0["projmatrix"].reshape(4, 4), atol=2e-6)
