"""ctypes loader of ``libr2s_hip.so`` (the C ABI declared in ``include/r2s_raster.h`` and
``include/r2s_physics.h``).

torch is imported first so that this library binds to the HIP runtime torch already loaded
(same SONAME ``libamdhip64.so.7``): device pointers and streams are then shared.
There is NO fallback: if the library is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("R2S_HIP_LIB", os.path.join(_PKG_ROOT, "libr2s_hip.so"))  # override: kernel experiments only

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class R2SGaussianSet(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("scale_modifier", C.c_float),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p),
        ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
    ]


class R2SRasterFrame(C.Structure):
    _fields_ = [
        ("set", C.c_int32), ("prefiltered", C.c_int32),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("z_threshold", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p), ("background", C.c_void_p),
        ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("radii", C.c_void_p),
    ]


class R2SRasterDebug(C.Structure):
    _fields_ = [
        ("total_gaussians", C.c_int64), ("num_rendered", C.c_int64),
        ("depths", C.c_void_p), ("radii", C.c_void_p), ("geom", C.c_void_p), ("tiles_touched", C.c_void_p),
        ("point_offsets", C.c_void_p), ("keys_sorted", C.c_void_p), ("point_list", C.c_void_p), ("ranges", C.c_void_p),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.r2s_last_error.restype = C.c_char_p
    L.r2s_version.restype = C.c_int
    L.r2s_raster_forward.restype = i64
    L.r2s_raster_forward.argtypes = [
        ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp,
        vp, f32, f32, i32, f32, vp, vp, vp, vp,
    ]
    L.r2s_raster_ctx_create.restype = i32
    L.r2s_raster_ctx_create.argtypes = [C.POINTER(vp)]
    L.r2s_raster_ctx_destroy.restype = None
    L.r2s_raster_ctx_destroy.argtypes = [vp]
    L.r2s_raster_ctx_scratch_bytes.restype = C.c_size_t
    L.r2s_raster_ctx_scratch_bytes.argtypes = [vp]
    L.r2s_raster_forward_batch.restype = i64
    L.r2s_raster_forward_batch.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp]
    L.r2s_raster_ctx_set_timing.restype = None
    L.r2s_raster_ctx_set_timing.argtypes = [vp, i32]
    L.r2s_raster_ctx_set_async.restype = None
    L.r2s_raster_ctx_set_async.argtypes = [vp, i32]
    L.r2s_raster_ctx_poll.restype = i32
    L.r2s_raster_ctx_poll.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(C.c_int32)]
    L.r2s_raster_ctx_set_tile_culling.restype = None
    L.r2s_raster_ctx_set_tile_culling.argtypes = [vp, i32]
    L.r2s_raster_ctx_stage_ms.restype = f32
    L.r2s_raster_ctx_stage_ms.argtypes = [vp, i32]
    L.r2s_raster_ctx_set_aux.restype = None
    L.r2s_raster_ctx_set_aux.argtypes = [vp, vp, vp]
    L.r2s_raster_ctx_debug.restype = i32
    L.r2s_raster_ctx_debug.argtypes = [vp, C.POINTER(R2SRasterDebug)]
    L.r2s_memcpy_d2d.restype = i32
    L.r2s_memcpy_d2d.argtypes = [vp, vp, C.c_size_t, vp]
    _lib = L
    return L


def kernel_source_sha16() -> str:
    """sha256[:16] over the kernel sources and C headers the library is built from (csrc/*.hip, csrc/*.h, include/*.h, sorted by
    name).  Counter summaries under profiles/ carry it, so that bench.py can tell a summary collected on THESE kernels from a
    stale one (the GPU box has no .git: a commit id cannot be read there)."""
    import glob
    import hashlib

    root = os.path.dirname(_PKG_ROOT)
    files = sorted(glob.glob(os.path.join(_PKG_ROOT, "csrc", "*.hip")) + glob.glob(os.path.join(_PKG_ROOT, "csrc", "*.h")) + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class R2SError(RuntimeError):
    pass


_ERR = {-1: "invalid argument", -2: "HIP runtime error", -3: "scratch allocation failed",
        -4: "Point is filtered although prefiltered is set. This shouldn't happen!",
        -5: "more than 2^32-1 Gaussian/tile instances"}


def check(rc: int, what: str, reason: bool = False) -> int:
    """``reason``: the entry point clears the library's message on entry, so an 'invalid' return with a message says why."""
    if rc < 0:
        detail = lib().r2s_last_error().decode() if rc == -2 or (reason and rc == -1) else ""
        raise R2SError(f"{what}: {_ERR.get(int(rc), 'error %d' % rc)} {detail}".strip())
    return int(rc)


def cur_stream(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
