"""Where the observation sink spends its time: D2H into the shared ring, JPEG encode, worker round trip."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import numpy as np, torch
from r2s_hip.sink import ObservationSink, pack_u8
from r2s_hip import _sink_worker

def main():
  E, V, H, W = 32, 2, 480, 640
  img = torch.rand(E, V, 3, H, W, device="cuda")
  x = torch.rand(E, 15000, 3, device="cuda")
  sink = ObservationSink("/tmp/sink_diag", E, V, H, W, slots=6, state_bytes=2 * E * 15000 * 12 + 4096)
  print("registered", sink._registered, "workers", len(sink._procs), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
  torch.cuda.synchronize()
  for rep in range(3):
      t0 = time.perf_counter()
      sink.submit(rep, img, state=dict(x=x, v=x))
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      print(f"submit {1e3*(t1-t0):.2f} ms, gpu+copy done after {1e3*(t2-t0):.2f} ms")
  t0 = time.perf_counter()
  sink.close()
  print(f"drain {time.perf_counter()-t0:.3f} s, frames {sink.frames_written}")
  px = (np.random.rand(H, W, 3) * 255).astype(np.uint8)
  t0 = time.perf_counter()
  for _ in range(10):
      b = _sink_worker.jpeg_bytes(px)
  print(f"jpeg encode {1e2*(time.perf_counter()-t0):.2f} ms per frame, {len(b)} bytes (noise image)")


if __name__ == '__main__':
    main()
