"""Observation sink: what ``experiments/eval_policy.py:145-213, 240-267`` writes per env step — one
JPEG per camera (``episode_XXXX/camera_C/rgb/NNNNNN.jpg``, plus start / final images), the robot record
(``robot/NNNNNN.json``) and the state pickle (``state/NNNNNN.pkl``) — for a batch of environments, OFF the step's critical path:

  device   r2s_obs_pack_u8: all frames of the batch -> interleaved uint8 pixels in one kernel (clamp, * 255, truncate: the
           reference's ``(image.cpu().numpy().transpose(1, 2, 0) * 255).astype(np.uint8)`` after its clamp, bit for bit)
  copy     one asynchronous D2H of the packed pixels (and of the state tensors) into a ring slot of page-locked SHARED memory,
           fenced by an event
  host     a dispatcher thread waits for the slot's event and hands one job per environment to a pool of worker PROCESSES
           (r2s_hip/_sink_worker.py: PIL JPEG at cv2's default quality 95, or uncompressed BMP; cv2 itself is not in this
           image), which read the slot in place and write the files; the slot is freed when its jobs are done.  Processes, not
           threads: 64 JPEG encodes per step on threads starve the rollout's own Python thread of the GIL (measured: 182 ms
           per step instead of 24).

``submit`` only enqueues GPU work; it blocks the caller only when every ring slot is still being written (counted in
``stalls``).  ``workers=0`` writes from the dispatcher thread itself (tests, tiny batches).  The reference does all of this
synchronously inside the episode loop."""
from __future__ import annotations

import ctypes as C
import multiprocessing as mp
import os
import queue
import threading
from multiprocessing import shared_memory
from typing import Optional

import numpy as np
import torch

from . import _lib, _sink_worker
from ._lib import check, cur_stream
from ._sink_worker import bmp_bytes  # noqa: F401  (re-exported)

_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        L.r2s_obs_pack_u8.restype = C.c_int
        L.r2s_obs_pack_u8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        _bound = True
    return L


def pack_u8(color: torch.Tensor, bgr: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """color [..., 3, H, W] float32 (device) -> uint8 [..., H, W, 3] (BGR like cv2.cvtColor(..., COLOR_RGB2BGR), or RGB)."""
    assert color.is_cuda and color.dtype == torch.float32 and color.shape[-3] == 3
    c = color.contiguous()
    H, W = int(c.shape[-2]), int(c.shape[-1])
    F = int(c.numel() // (3 * H * W))
    if out is None:
        out = torch.empty(*c.shape[:-3], H, W, 3, dtype=torch.uint8, device=c.device)
    assert out.is_contiguous() and out.numel() == F * H * W * 3
    with torch.cuda.device(c.device):
        check(_bind().r2s_obs_pack_u8(c.data_ptr(), F, H, W, int(bool(bgr)), out.data_ptr(), cur_stream(c.device)), "r2s_obs_pack_u8")
    return out


def default_format() -> str:
    try:
        import PIL  # noqa: F401
        return "jpg"
    except Exception:
        return "bmp"


class ObservationSink:
    def __init__(self, out_dir: str, n_env: int, n_cam: int, height: int, width: int, device="cuda:0", slots: int = 4, run_name: str = "run",
                 fmt: Optional[str] = None, write_images: bool = True, episode_ids=None, workers: Optional[int] = None, state_bytes: int = 0):
        self.root = os.path.join(out_dir, run_name)
        self.E, self.V, self.H, self.W = int(n_env), int(n_cam), int(height), int(width)
        self.device = torch.device(device)
        self.episode_ids = list(episode_ids) if episode_ids is not None else list(range(self.E))
        self.write_images = write_images
        self.ext = fmt or default_format()
        self.slots = int(slots)
        self._px_bytes = self.E * self.V * self.H * self.W * 3
        self._state_cap = int(state_bytes)
        self._slot_bytes = (self._px_bytes + self._state_cap + 4095) // 4096 * 4096
        # ring in shared memory, page-locked so that the D2H copies are asynchronous
        self._shm = shared_memory.SharedMemory(create=True, size=self._slot_bytes * self.slots)
        self._host = torch.frombuffer(self._shm.buf, dtype=torch.uint8)
        self._registered = False
        try:
            rc = torch.cuda.cudart().cudaHostRegister(self._host.data_ptr(), self._host.numel(), 0)
            self._registered = int(rc) == 0
        except Exception:
            self._registered = False   # copies still work, they just synchronise
        self._dev = [torch.empty(self.E, self.V, self.H, self.W, 3, dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
        # Round 6: the D2H copies (59 MB of frames + the state per 32-environment step) run on a stream of their own.  On the rollout's stream
        # they sat between two env steps — the next step's graphs waited 2 ms for a DMA they have nothing to do with (sustained episodes ran at
        # 0.89 of the closed-loop figure).  The state is snapshotted into the slot's device staging on the rollout's stream first (the next
        # step overwrites it), then frames and state leave from the staging.
        self._dev_state = [torch.empty(max(self._slot_bytes - self._px_bytes, 1), dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._free: "queue.Queue[int]" = queue.Queue()
        for k in range(self.slots):
            self._free.put(k)
        self._work: "queue.Queue" = queue.Queue()
        self.stalls, self.frames_written, self.steps_written, self.videos_written = 0, 0, 0, 0
        self._err = None
        for e in self.episode_ids:
            for c in range(self.V):
                os.makedirs(os.path.join(self.root, f"episode_{e:04d}", f"camera_{c}", "rgb"), exist_ok=True)
            os.makedirs(os.path.join(self.root, f"episode_{e:04d}", "robot"), exist_ok=True)
            os.makedirs(os.path.join(self.root, f"episode_{e:04d}", "state"), exist_ok=True)
        os.makedirs(os.path.join(self.root, "start_images"), exist_ok=True)
        os.makedirs(os.path.join(self.root, "final_images"), exist_ok=True)
        n_workers = max(0, min(self.E, (os.cpu_count() or 2) - 2, 32)) if workers is None else int(workers)
        self._procs, self._job_q, self._done_q = [], None, None
        if n_workers > 0:
            ctx = mp.get_context("spawn")
            self._job_q, self._done_q = ctx.Queue(), ctx.Queue()
            for _ in range(n_workers):
                p = ctx.Process(target=_sink_worker.worker_main, args=(self._job_q, self._done_q, self._shm.name, self.ext), daemon=True)
                p.start()
                self._procs.append(p)
            for _ in self._procs:   # block until every worker has imported its encoder: the rollout must not race their start-up
                self._done_q.get(timeout=120)
        self._thread = threading.Thread(target=self._dispatch, daemon=True)
        self._thread.start()

    # ---- producer side (the rollout loop) ------------------------------------------------------------------------------
    def submit(self, cnt: int, color: torch.Tensor, state: Optional[dict] = None, robot: Optional[list] = None, final: bool = False,
               ready: Optional[torch.cuda.Event] = None):
        """color: [n_env, n_cam, 3, H, W] float32 on the device (the rasteriser's output, unclamped); state: dict of device
        tensors with a leading n_env axis (e.g. x, v: what ``env.get_state()['renderer']`` holds; needs ``state_bytes`` >= their
        total size at construction); robot: per-env dicts for ``robot/NNNNNN.json``.  Returns immediately unless every ring
        slot is still being written.  ``ready``: an event after which ``color`` is complete when the frames were produced on
        ANOTHER stream than the current one (BatchedRollout.set_pipelined renders on a private stream: pass its
        ``_render_done`` event, or call ``wait_render()`` first); the pack kernel waits for it on the device."""
        if self._err:
            raise self._err
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        try:
            k = self._free.get_nowait()
        except queue.Empty:
            self.stalls += 1
            k = self._free.get()
        base = k * self._slot_bytes
        pack_u8(color.reshape(self.E, self.V, 3, self.H, self.W), bgr=True, out=self._dev[k])
        desc, off, staged = [], base + self._px_bytes, []
        for name, t in (state or {}).items():
            t = t.detach().contiguous()
            nbytes = t.numel() * t.element_size()
            if off + nbytes > base + self._slot_bytes:
                raise ValueError("state does not fit the ring slot: pass state_bytes >= the total size of the state tensors")
            so = off - (base + self._px_bytes)
            self._dev_state[k][so: so + nbytes].copy_(t.reshape(-1).view(torch.uint8))   # D2D on the rollout's stream: a snapshot
            staged.append((so, off, nbytes))
            desc.append((name, str(t.dtype).replace("torch.", ""), tuple(t.shape), off))
            off = (off + nbytes + 63) // 64 * 64
        main = torch.cuda.current_stream(self.device)
        cs = self._copy_stream if (self._copy_stream is not None and self._registered) else main
        if cs is not main:
            cs.wait_stream(main)
        with torch.cuda.stream(cs):
            self._host[base: base + self._px_bytes].copy_(self._dev[k].reshape(-1), non_blocking=self._registered)
            for so, ho, nbytes in staged:
                self._host[ho: ho + nbytes].copy_(self._dev_state[k][so: so + nbytes], non_blocking=self._registered)
            ev = torch.cuda.Event()
            ev.record(cs)
        self._work.put((k, ev, int(cnt), desc, robot, bool(final)))

    def make_videos(self, frame_rate: int = 10):
        """``vis_camera_C.mp4`` per episode and camera from the frames written so far — eval_policy.py:261-267 -> make_video
        (experiments/utils/ffmpeg.py:5-21: ffmpeg, libx264, yuv420p), run by the worker processes after the frame queue has
        drained.  Returns the number of videos written (0 when ffmpeg is not installed, or with the BMP fallback format)."""
        if self._err:
            raise self._err
        self._videos_done = threading.Event()          # before the work item: the dispatcher may finish (BMP fallback: at once) and set() it
        self._work.put(("videos", int(frame_rate)))
        while not self._videos_done.wait(0.5):
            if self._err or not self._thread.is_alive():     # a dispatcher that died never sets the event
                break
        if self._err:
            raise self._err
        return self.videos_written

    def close(self):
        """Drain the queue, stop the dispatcher and the workers, release the ring."""
        self._work.put(None)
        self._thread.join()
        for _ in self._procs:
            self._job_q.put(None)
        for p in self._procs:
            p.join(30)
        if self._registered:
            try:
                torch.cuda.synchronize(self.device)
                torch.cuda.cudart().cudaHostUnregister(self._host.data_ptr())
            except Exception:
                pass
        del self._host
        try:
            self._shm.close()
            self._shm.unlink()
        except Exception:
            pass
        if self._err:
            raise self._err

    # ---- consumer side ---------------------------------------------------------------------------------------------------
    def _dispatch(self):
        try:
            jid = 0
            while True:
                item = self._work.get()
                if item is None:
                    return
                if item[0] == "videos":   # every frame submitted before this point has been written (the queue is FIFO)
                    try:
                        self.videos_written = self._write_videos(item[1], jid)
                        jid += len(self.episode_ids) * self.V
                    finally:
                        self._videos_done.set()
                    continue
                k, ev, cnt, desc, robot, final = item
                ev.synchronize()
                base = k * self._slot_bytes
                per_env_px = self.V * self.H * self.W * 3
                jobs = []
                for i, e in enumerate(self.episode_ids):
                    sd = []
                    for name, dt, shape, off in desc:   # environment i's slice of every state tensor
                        row = int(np.prod(shape[1:])) * np.dtype(dt).itemsize
                        sd.append((name, dt, tuple(shape[1:]), off + i * row))
                    jobs.append((jid, base + i * per_env_px, self.V, self.H, self.W, sd, self.root, e, cnt, final, self.write_images,
                                 robot[i] if robot is not None else None))
                    jid += 1
                if self._procs:
                    for j in jobs:
                        self._job_q.put(j)
                    got = 0
                    while got < len(jobs):
                        try:
                            _, n, err = self._done_q.get(timeout=1.0)
                        except queue.Empty:   # a dead worker must not hang the rollout
                            if not all(p.is_alive() for p in self._procs):
                                raise RuntimeError("an observation-sink worker process died (spawned workers re-import __main__: "
                                                   "the calling script needs an `if __name__ == '__main__':` guard)")
                            continue
                        if err:
                            raise RuntimeError(f"sink worker: {err}")
                        self.frames_written += n
                        got += 1
                else:
                    buf = np.ndarray((self._shm.size,), np.uint8, buffer=self._shm.buf)
                    for (_, px_off, V, H, W, sd, root, e, c, fin, wi, rb) in jobs:
                        px = buf[px_off: px_off + V * H * W * 3].reshape(V, H, W, 3)
                        st = {name: np.ndarray(shape, np.dtype(dt), buffer=self._shm.buf, offset=off) for name, dt, shape, off in sd} or None
                        self.frames_written += _sink_worker.write_env(px, st, root, self.ext, _sink_worker.ENCODERS[self.ext], e, c, fin, wi, rb)
                self.steps_written += 1
                self._free.put(k)
        except Exception as e:  # surfaced by the next submit / close
            self._err = e
            self._free.put(0)

    def _write_videos(self, frame_rate, jid0):
        if self.ext != "jpg" or not self.write_images:
            return 0
        jobs = []
        for e in self.episode_ids:
            ep = os.path.join(self.root, f"episode_{e:04d}")
            for c in range(self.V):
                jobs.append(("video", jid0 + len(jobs), os.path.join(ep, f"camera_{c}", "rgb"), os.path.join(ep, f"vis_camera_{c}.mp4"), "%06d.jpg", frame_rate))
        n = 0
        if self._procs:
            for j in jobs:
                self._job_q.put(j)
            for _ in jobs:
                _, ok, err = self._done_q.get(timeout=600)
                if err:
                    raise RuntimeError(f"sink worker (video): {err}")
                n += ok
        else:
            for j in jobs:
                n += int(_sink_worker.make_video(*j[2:]))
        return n
