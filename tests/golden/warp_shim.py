"""A minimal float32 interpreter shim for the subset of the `warp` API that the arithmetic kernels of the reference's
sim/physics/spring_mass_warp.py use, so that THOSE KERNEL BODIES — the reference's own Python source, imported, not
copied — can be executed thread by thread on the CPU to produce golden fixtures (tests/golden/make_physics_golden.py).

This is not a warp implementation.  What it provides: float32 vec3 values (numpy), length / dot / cross / normalize / min /
max / clamp / exp with float32 rounding after every operation (no FMA), tid(), atomic_add / atomic_sub applied in thread
order, no-op decorators and type annotations.  What it deliberately does NOT emulate: HashGrid and Mesh/BVH queries —
`mesh_query_point_sign_winding_number` is answered by a callable the caller attaches to the mesh handle (the oracle's
restated closest-point / winding-number routine), so fixtures for `mesh_collision` pin the kernel's response arithmetic
but not the query; the two hash-grid kernels are not executed at all."""
import types

import numpy as np

f32 = np.float32
_tid = [0]


def _make():
    wp = types.ModuleType("warp")
    wp.float32, wp.int32, wp.uint64, wp.vec2i = np.float32, np.int32, np.uint64, None
    wp.bool = bool
    wp.array = wp.array2d = lambda *a, **k: None            # annotations only
    wp.kernel = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda fn: fn))
    wp.func = lambda fn: fn
    wp.tid = lambda: _tid[0]
    wp.vec3 = lambda x=0.0, y=0.0, z=0.0: np.array([x, y, z], f32)
    wp.exp = lambda x: np.exp(f32(x))
    wp.max = lambda a, b: a if a >= b else b
    wp.min = lambda a, b: a if a <= b else b
    wp.clamp = lambda x, low, high: f32(min(max(f32(x), f32(low)), f32(high)))

    def dot(a, b):
        return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))

    def length(a):
        return f32(np.sqrt(dot(a, a)))

    def cross(a, b):
        return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0])], f32)

    def normalize(a):
        l = length(a)
        return (a / l).astype(f32) if l > 0 else np.zeros(3, f32)

    def atomic_add(arr, idx, val):
        arr[idx] = arr[idx] + val

    def atomic_sub(arr, idx, val):
        arr[idx] = arr[idx] - val

    wp.dot, wp.length, wp.cross, wp.normalize, wp.atomic_add, wp.atomic_sub = dot, length, cross, normalize, atomic_add, atomic_sub

    def mesh_query_point_sign_winding_number(mesh, p, max_dist, accuracy=3.0, threshold=0.6):
        return mesh.query(np.asarray(p, f32), max_dist, threshold)

    wp.mesh_query_point_sign_winding_number = mesh_query_point_sign_winding_number
    wp.mesh_eval_position = lambda mesh, face, u, v: np.asarray(mesh.last_point, f32)
    return wp


def launch(kernel, dim, inputs):
    """wp.launch for 1-D grids: threads run one after the other in tid order."""
    for t in range(int(dim)):
        _tid[0] = t
        kernel(*inputs)


warp = _make()
