"""Observation sink (row f4): the packing kernel reproduces the reference's host conversion bit for bit, and the asynchronous
writer produces the reference's file layout without blocking the producer."""
import json
import os
import pickle as pkl

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference_conversion(image_chw):
    """experiments/eval_policy.py:157-158 after gs_renderer.py:949, with cv2.cvtColor(RGB2BGR) written as a channel flip."""
    clamped = np.clip(image_chw, 0.0, 1.0).astype(np.float32)
    return (clamped.transpose(1, 2, 0) * 255).astype(np.uint8)[:, :, ::-1]


@pytest.mark.parametrize("shape", [(3, 2, 48, 64), (1, 1, 37, 53), (2, 4, 30, 41)])
def test_pack_u8_equals_the_reference_host_conversion(shape):
    import torch
    from r2s_hip.sink import pack_u8

    rng = np.random.default_rng(sum(shape))
    E, V, H, W = shape
    img = rng.uniform(-0.2, 1.2, (E, V, 3, H, W)).astype(np.float32)
    img[0, 0, :, 0, :8] = [[0.0, 1.0, 0.5, 254.9999 / 255, 1.0 / 255, 0.999999, 1e-8, 0.00392157]] * 3   # edge values of the truncation
    out = pack_u8(torch.from_numpy(img).cuda(), bgr=True).cpu().numpy()
    assert out.shape == (E, V, H, W, 3) and out.dtype == np.uint8
    for e in range(E):
        for v in range(V):
            assert np.array_equal(out[e, v], _reference_conversion(img[e, v])), (e, v)
    rgb = pack_u8(torch.from_numpy(img).cuda(), bgr=False).cpu().numpy()
    assert np.array_equal(rgb[..., ::-1], out)


@pytest.mark.parametrize("workers", [0, 2], ids=["dispatcher thread writes", "worker processes write"])
def test_sink_writes_the_reference_layout_asynchronously(tmp_path, workers):
    import torch
    from r2s_hip.sink import ObservationSink, bmp_bytes

    E, V, H, W = 3, 2, 24, 32
    sink = ObservationSink(str(tmp_path), E, V, H, W, slots=2, run_name="demo", fmt="bmp", episode_ids=[7, 8, 9], workers=workers, state_bytes=1 << 16)
    rng = np.random.default_rng(0)
    frames, states = [], []
    for cnt in range(5):
        img = torch.from_numpy(rng.uniform(-0.1, 1.1, (E, V, 3, H, W)).astype(np.float32)).cuda()
        x = torch.from_numpy(rng.normal(size=(E, 10, 3)).astype(np.float32)).cuda()
        frames.append(img.cpu().numpy()); states.append(x.cpu().numpy())
        sink.submit(cnt, img, state=dict(x=x), robot=[{"obs.ee_pos": [float(cnt), float(e), 0.0]} for e in range(E)], final=(cnt == 4))
    sink.close()
    assert sink.steps_written == 5 and sink.frames_written == 5 * E * V
    root = tmp_path / "demo"
    for cnt in range(5):
        for i, e in enumerate((7, 8, 9)):
            for c in range(V):
                raw = (root / f"episode_{e:04d}" / f"camera_{c}" / "rgb" / f"{cnt:06d}.bmp").read_bytes()
                assert raw == bmp_bytes(_reference_conversion(frames[cnt][i, c]))
            st = pkl.load(open(root / f"episode_{e:04d}" / "state" / f"{cnt:06d}.pkl", "rb"))
            assert np.array_equal(st["renderer"]["x"], states[cnt][i])
            rb = json.load(open(root / f"episode_{e:04d}" / "robot" / f"{cnt:06d}.json"))
            assert rb["obs.ee_pos"][:2] == [float(cnt), float(i)]
    assert (root / "start_images" / "episode_0007_camera_0.bmp").read_bytes() == bmp_bytes(_reference_conversion(frames[0][0, 0]))
    assert (root / "final_images" / "episode_0009_camera_1.bmp").read_bytes() == bmp_bytes(_reference_conversion(frames[4][2, 1]))


def test_default_encoder_is_jpeg_when_pil_is_present(tmp_path):
    import torch
    from r2s_hip.sink import ObservationSink, default_format

    if default_format() != "jpg":
        pytest.skip("PIL not installed: BMP fallback")
    from PIL import Image

    sink = ObservationSink(str(tmp_path), 1, 1, 48, 64, run_name="j", workers=1)
    img = torch.zeros(1, 1, 3, 48, 64, device="cuda"); img[:, :, 0] = 1.0      # pure red
    sink.submit(0, img)
    sink.close()
    im = np.asarray(Image.open(tmp_path / "j" / "episode_0000" / "camera_0" / "rgb" / "000000.jpg"))
    assert im.shape == (48, 64, 3) and im[..., 0].mean() > 240 and im[..., 1].mean() < 15 and im[..., 2].mean() < 15   # stored as RGB red


@pytest.mark.parametrize("workers", [0, 2])
def test_episode_videos_are_made_after_the_frames_have_drained(tmp_path, workers, monkeypatch):
    """eval_policy.py:261-267: one vis_camera_C.mp4 per episode and camera through ffmpeg (experiments/utils/ffmpeg.py).  A
    stand-in ffmpeg on PATH records how many frames exist when it is invoked: every submitted frame must have been written."""
    import stat

    import torch
    from r2s_hip.sink import ObservationSink, default_format

    if default_format() != "jpg":
        pytest.skip("PIL not installed: BMP fallback has no video step")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    fake = bindir / "ffmpeg"
    fake.write_text("#!/bin/sh\nfor a in \"$@\"; do last=\"$a\"; case \"$prev\" in -i) src=\"$a\";; esac; prev=\"$a\"; done\nls \"$(dirname \"$src\")\" | wc -l > \"$last\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{bindir}:{os.environ['PATH']}")
    E, V = 2, 2
    sink = ObservationSink(str(tmp_path), E, V, 24, 32, run_name="v", workers=workers, slots=2)
    for cnt in range(6):
        sink.submit(cnt, torch.rand(E, V, 3, 24, 32, device="cuda"))
    assert sink.make_videos(frame_rate=10) == E * V
    sink.close()
    for e in range(E):
        for c in range(V):
            assert int((tmp_path / "v" / f"episode_{e:04d}" / f"vis_camera_{c}.mp4").read_text()) == 6
