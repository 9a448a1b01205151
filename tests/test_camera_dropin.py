"""R0 (`setup_camera`, sim/utils/gs/transform_utils.py:7-31): the DROP-IN function itself — not only the oracle's restatement —
against the fixture the reference's own function produced (tests/golden/camera_side_848x480.json <- make_camera_golden.py).
On the CPU the drop-in issues the reference's torch operations one for one, so every field is bit-equal; the oracle's numpy
restatement uses a different 4x4 inverse / matmul and is held to 2 ulps.  (The `device='cuda'` run is in
test_parity_round2_gpu.py.)"""
import json
import os

import numpy as np

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "camera_side_848x480.json")))


def _ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(-1).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).reshape(-1).view(np.int32).astype(np.int64)
    return int(np.abs(a - b).max())


def test_dropin_setup_camera_is_bit_equal_to_the_reference_fixture_on_cpu():
    from r2s_hip import synth
    from sim.utils.gs.transform_utils import setup_camera

    cam = setup_camera(G["w"], G["h"], G["K"], np.linalg.inv(np.array(G["c2w"])), near=0.01, far=100.0, device="cpu")
    assert np.array_equal(cam.viewmatrix.numpy().reshape(-1), np.asarray(G["viewmatrix"], np.float32))
    assert np.array_equal(cam.projmatrix.numpy().reshape(-1), np.asarray(G["projmatrix"], np.float32))
    assert np.array_equal(cam.campos.numpy(), np.asarray(G["campos"], np.float32))
    assert np.array_equal(cam.bg.numpy(), np.asarray(G["bg"], np.float32))
    assert (cam.tanfovx, cam.tanfovy, cam.z_threshold, cam.scale_modifier, cam.sh_degree, cam.prefiltered) == \
           (G["tanfovx"], G["tanfovy"], G["z_threshold"], G["scale_modifier"], G["sh_degree"], G["prefiltered"])
    assert (cam.image_height, cam.image_width) == (G["h"], G["w"])
    assert np.allclose(np.array(G["K"]), synth.SIDE_K) and np.allclose(np.array(G["c2w"]), synth.SIDE_C2W)  # the fixture is the yaml's camera


def test_oracle_setup_camera_within_two_ulps_of_the_reference_fixture():
    cam = oracle.setup_camera(G["w"], G["h"], G["K"], np.linalg.inv(np.array(G["c2w"])), z_threshold=G["z_threshold"])
    assert cam["tanfovx"] == G["tanfovx"] and cam["tanfovy"] == G["tanfovy"]
    assert np.array_equal(np.asarray(cam["viewmatrix"], np.float32).reshape(-1), np.asarray(G["viewmatrix"], np.float32))
    assert _ulps(cam["projmatrix"], G["projmatrix"]) <= 2 and _ulps(cam["campos"], G["campos"]) <= 2
