"""Worker process of the observation sink (r2s_hip/sink.py): encodes and writes the frames / state of one environment of one
ring slot.  Imports neither torch nor the HIP library — it only sees the shared-memory ring the main process copies into."""
import io
import json
import os
import pickle as pkl
import struct

import numpy as np


def bmp_bytes(bgr: np.ndarray) -> bytes:
    """24-bit uncompressed BMP of an [H, W, 3] BGR image (BMP's native channel order; rows bottom-up, padded to 4 bytes)."""
    h, w, _ = bgr.shape
    row = (3 * w + 3) // 4 * 4
    body = np.zeros((h, row), np.uint8)
    body[:, : 3 * w] = bgr[::-1].reshape(h, 3 * w)
    head = b"BM" + struct.pack("<IHHI", 54 + row * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, row * h, 2835, 2835, 0, 0)
    return head + body.tobytes()


def jpeg_bytes(bgr: np.ndarray) -> bytes:
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(buf, "JPEG", quality=95)   # cv2.imwrite's default JPEG quality
    return buf.getvalue()


ENCODERS = {"jpg": jpeg_bytes, "bmp": bmp_bytes}


def write_env(px, state, root, ext, encode, e, cnt, final, write_images, robot):
    """px: [V, H, W, 3] uint8 BGR of one environment; state: {name: array} of that environment (or None)."""
    ep = os.path.join(root, f"episode_{e:04d}")
    n = 0
    if write_images:
        for c in range(px.shape[0]):
            data = encode(px[c])
            with open(os.path.join(ep, f"camera_{c}", "rgb", f"{cnt:06d}.{ext}"), "wb") as f:
                f.write(data)
            if cnt == 0 or final:   # eval_policy.py:162-163, :253
                with open(os.path.join(root, "start_images" if cnt == 0 else "final_images", f"episode_{e:04d}_camera_{c}.{ext}"), "wb") as f:
                    f.write(data)
            n += 1
    if state:
        with open(os.path.join(ep, "state", f"{cnt:06d}.pkl"), "wb") as f:
            pkl.dump({"renderer": {k: np.array(v) for k, v in state.items()}}, f)
    if robot is not None:
        with open(os.path.join(ep, "robot", f"{cnt:06d}.json"), "w") as f:
            json.dump(robot, f, indent=4)
    return n


def make_video(image_root, video_path, image_pattern="%06d.jpg", frame_rate=10, ffmpeg="ffmpeg"):
    """The episode video of one camera: the reference's ``make_video`` (experiments/utils/ffmpeg.py:5-21, called per camera at the
    end of an episode, experiments/eval_policy.py:261-267) — the same ``ffmpeg`` command line (libx264, yuv420p) over the frames
    the sink wrote.  ffmpeg is an external program, as it is for the reference; returns False when it is not installed."""
    import shutil
    import subprocess

    exe = shutil.which(ffmpeg)
    if exe is None:
        return False
    r = subprocess.run([exe, "-y", "-hide_banner", "-loglevel", "error", "-framerate", str(frame_rate), "-i", os.path.join(str(image_root), image_pattern),
                        "-c:v", "libx264", "-pix_fmt", "yuv420p", str(video_path)])
    return r.returncode == 0


def worker_main(job_q, done_q, shm_name, fmt):
    from multiprocessing import shared_memory

    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        buf = np.ndarray((shm.size,), np.uint8, buffer=shm.buf)
        encode = ENCODERS[fmt]
        encode(np.zeros((8, 8, 3), np.uint8))     # import the encoder now, not on the first frame
        done_q.put((-1, 0, None))                  # ready
        while True:
            job = job_q.get()
            if job is None:
                return
            try:
                if job[0] == "video":   # ("video", jid, image_root, video_path, pattern, frame_rate)
                    done_q.put((job[1], int(make_video(*job[2:])), None))
                    continue
                (jid, px_off, V, H, W, state_desc, root, e, cnt, final, write_images, robot) = job
                px = buf[px_off: px_off + V * H * W * 3].reshape(V, H, W, 3)
                state = {name: np.ndarray(shape, np.dtype(dt), buffer=shm.buf, offset=off) for name, dt, shape, off in state_desc} or None
                done_q.put((jid, write_env(px, state, root, fmt, encode, e, cnt, final, write_images, robot), None))
            except Exception as ex:  # reported to the main process, which raises it from submit() / close()
                done_q.put((job[1] if job[0] == "video" else job[0], 0, repr(ex)))
    finally:
        shm.close()
