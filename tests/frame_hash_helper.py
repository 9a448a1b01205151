"""Helper of tests/test_raster_gpu.py::test_inline_assembly_blend_equals_its_cxx_specification_bit_for_bit: renders a fixed set of
frames through whichever build of the library R2S_HIP_LIB names and prints one sha256 per frame set (colour + depth + n_contrib-free
outputs).  Run as a subprocess: the library is bound at import."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "real2sim-eval_amd")]


def main():
    import torch
    from r2s_hip.rollout import BatchedRollout
    from util_raster import hip_render, scene_and_camera

    for tag, (P, W, H, seed, cam) in dict(side=(40000, 640, 480, 31, "side"), wrist=(40000, 640, 480, 31, "wrist"), ref_frame=(30000, 848, 480, 7, "side"),
                                          ragged=(5000, 333, 217, 3, "wrist")).items():
        sc, c = scene_and_camera(P, W, H, seed, cam=cam, bg=(0.1, 0.2, 0.3))
        col, radii, dep = hip_render(sc, c)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(col).tobytes()); h.update(np.ascontiguousarray(dep).tobytes()); h.update(np.ascontiguousarray(radii).tobytes())
        print("HASH", tag, h.hexdigest(), flush=True)
    ro = BatchedRollout("tiny", num_substeps=20, seed=5, n_env=3)          # the batched entry point, per-environment wrist cameras
    h = hashlib.sha256()
    for _ in range(3):
        ro.step()
        col, dep = ro.observations()
        torch.cuda.synchronize()
        h.update(col.cpu().numpy().tobytes()); h.update(dep.cpu().numpy().tobytes())
    print("HASH", "rollout", h.hexdigest(), flush=True)


if __name__ == "__main__":
    main()
