"""The C-ABI library loads (no GPU needed) and exports every entry point include/*.h declares; the Python
structures mirror the C layouts; the product package never imports the oracle."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "real2sim-eval_amd", "libr2s_hip.so")


def _declared():
    names = []
    for h in ("r2s_raster.h", "r2s_physics.h", "r2s_skinning.h", "r2s_metrics.h", "r2s_robot.h", "r2s_obs.h", "r2s_camera.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(r2s_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_fn") and n != "r2s_stream_t"))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build with: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exported = set(re.findall(r" T (r2s_[a-z0-9_]+)", out))
    declared = _declared()
    assert len(declared) >= 25
    missing = [n for n in declared if n not in exported]
    assert not missing, missing


def test_library_loads_and_reports_version():
    from r2s_hip import _lib

    L = _lib.lib()
    assert L.r2s_version() == 100
    assert L.r2s_last_error() == b""
    for n in _declared():
        getattr(L, n)


def test_struct_layouts_match_headers():
    from r2s_hip import _lib, physics

    # field counts / sizes computed by hand from include/r2s_raster.h and include/r2s_physics.h (LP64)
    assert ctypes.sizeof(_lib.R2SGaussianSet) == 4 * 4 + 7 * 8
    assert ctypes.sizeof(_lib.R2SRasterFrame) == 2 * 4 + 3 * 4 + 4 + 7 * 8  # 4 bytes of padding before the pointers
    assert ctypes.sizeof(_lib.R2SRasterDebug) == 2 * 8 + 8 * 8
    assert ctypes.sizeof(physics.R2SPhysParams) == 16 * 4
    assert ctypes.sizeof(physics.R2SFlavourIn) == 29 * 4 and ctypes.sizeof(physics.R2SFlavourOut) == 14 * 4 + 192   # include/r2s_physics.h: R2SFlavourIn / R2SFlavourOut
    assert ctypes.sizeof(physics.R2SPhysDesc) == 64 + 3 * 4 + 4 + 7 * 8 + 2 * 4 + 4 * 8 + 8


def test_invalid_arguments_are_rejected_without_a_gpu():
    from r2s_hip import _lib, physics

    L = physics._bind()
    d = physics.R2SPhysDesc()
    h = ctypes.c_void_p()
    assert L.r2s_phys_create(ctypes.byref(d), ctypes.byref(h), None) == -1  # R2S_ERR_INVALID: n_env == 0
    assert L.r2s_phys_step(None, 0, 0, None) == -1
    assert _lib.lib().r2s_raster_forward_batch(None, None, 0, None, 0, 64, 64, None, None) == -1
    from r2s_hip import skinning

    S = skinning._bind()
    assert S.r2s_skin_create(0, 8, None, 0, 16, None, None, ctypes.byref(h), None) == -1
    assert S.r2s_skin_interpolate_motions(None, 1, None, None, None, None, None) == -1
    from r2s_hip import metrics

    M = metrics._bind()
    assert M.r2s_metric_mse(0, 10, None, None, None, None) == -1
    assert M.r2s_metric_plane_crossings(1, 10, None, 5, None, None, None, 1e-12, None, None) == -1
    assert physics._bind().r2s_phys_set_eef_motion(None, None, None, None, None, None, None) == -1


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "real2sim-eval_amd")
    bad = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", s, flags=re.M) or "libr2s_oracle" in s:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_no_gpu_means_loud_failure_not_fallback():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    z = torch.zeros
    cam = GaussianRasterizationSettings(8, 8, 1.0, 1.0, z(3), 1.0, z(1, 4, 4), z(1, 4, 4), 0, z(3), False, 0.05)
    with pytest.raises(RuntimeError, match="no CPU path"):
        GaussianRasterizer(cam)(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), scales=z(2, 3), rotations=z(2, 4))
