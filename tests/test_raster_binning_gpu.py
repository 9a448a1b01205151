"""One-pass tile binning (csrc/raster.hip k_bin_*, round 6) against the two-pass radix sort it replaced and against the oracle.

The binning is a stable partition of the emitted (tile, Gaussian) instances by tile: its `point_list` and `ranges` must be the ones the
reference's single 64-bit sort produces (rasterizer_impl.cu:70-138, :306-321) — identical to the radix-sort path of rounds 1-5, which stays in
the library behind R2S_RASTER_RADIX_SORT (read at context creation) and for frames of more than 2 048 tiles.  Cases: every key width the
scatter kernel is instantiated for (9 / 10 / 11 bits), frames without a single instance inside a batch, chunks that end at frame
boundaries (frames far smaller and far larger than the 8 192-instance chunk), tile culling on and off, and the sync-free capacity path."""
import os

import numpy as np
import pytest

from util_raster import oracle_render

pytestmark = pytest.mark.gpu


def _ctx(dev, radix):
    from r2s_hip.raster import RasterBatch

    old = os.environ.get("R2S_RASTER_RADIX_SORT")
    if radix:
        os.environ["R2S_RASTER_RADIX_SORT"] = "1"
    else:
        os.environ.pop("R2S_RASTER_RADIX_SORT", None)
    try:
        return RasterBatch(dev)          # the knob is read once, here
    finally:
        if old is None:
            os.environ.pop("R2S_RASTER_RADIX_SORT", None)
        else:
            os.environ["R2S_RASTER_RADIX_SORT"] = old


def _render(rb, scenes, cams, frame_of, W, H, dev, cull):
    """frame_of: list of (scene index, camera index, z_threshold override or None).  Returns n, point_list, ranges, colour, depth."""
    import torch
    from r2s_hip.raster import _memcpy_d2d

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    rb.set_tile_culling(cull)
    sets = [rb.make_set(t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]), rotations=t(s["rotations"])) for s in scenes]
    F = len(frame_of)
    out_c = torch.empty(F, 3, H, W, device=dev); out_d = torch.empty(F, 1, H, W, device=dev)
    frames = []
    for f, (si, ci, zt) in enumerate(frame_of):
        c = cams[ci]
        frames.append(dict(set=si, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), campos=t(c["campos"]), bg=t(c["bg"]),
                           tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], z_threshold=c["z_threshold"] if zt is None else zt,
                           out_color=out_c[f], out_depth=out_d[f]))
    n = rb.forward(sets, frames, W, H)
    d = rb.debug()
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    r = torch.empty(F * tiles, 2, dtype=torch.int32, device=dev)
    _memcpy_d2d(r.data_ptr(), d["ranges_ptr"], r.numel() * 4, dev)
    torch.cuda.synchronize()
    return n, d["point_list"].numpy().astype(np.int64), r.cpu().numpy().astype(np.int64), out_c.cpu().numpy(), out_d.cpu().numpy()


@pytest.mark.parametrize("W,H,bits", [(320, 240, 9), (512, 384, 10), (640, 480, 11), (848, 480, 11)])
@pytest.mark.parametrize("cull", [False, True])
def test_binning_equals_radix_sort_path(W, H, bits, cull):
    """The same batch through both paths: instance count, point_list, ranges and every pixel identical.  The batch holds a frame whose
    Gaussians are all behind the near plane (no instance: its chunk count is zero), a 300-Gaussian frame (a fraction of one chunk) and
    frames of several chunks, side and wrist cameras."""
    import torch
    from r2s_hip import synth

    dev = torch.device("cuda:0")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    assert (tiles - 1).bit_length() == bits
    cams = [synth.side_camera(W, H), synth.wrist_camera(W, H)]
    scenes = [synth.gaussian_scene(9000, 11), synth.gaussian_scene(300, 12), synth.gaussian_scene(5000, 13)]
    frame_of = [(0, 0, None), (1, 0, None), (2, 1, None), (0, 1, 1.0e6), (2, 0, None), (1, 1, None)]  # frame 3: everything culled
    a = _render(_ctx(dev, False), scenes, cams, frame_of, W, H, dev, cull)
    b = _render(_ctx(dev, True), scenes, cams, frame_of, W, H, dev, cull)
    assert a[0] == b[0] and a[0] > 8192
    assert np.array_equal(a[1], b[1]), "point_list"
    assert np.array_equal(a[2], b[2]), "ranges"
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), "pixels"
    r = a[2].reshape(len(frame_of), tiles, 2)
    assert (r[3] == 0).all(), "a frame without instances has empty ranges"
    assert (r[..., 1] - r[..., 0]).sum() == a[0]


def test_binning_point_list_and_ranges_equal_the_oracle_batched():
    """Against the CPU oracle (the reference's duplicateWithKeys + 64-bit sort + identifyTileRanges), culling off: every frame's slice of
    the batch's point_list is the oracle's list of that frame (indices shifted by the frame's first Gaussian), ranges shifted by the
    instances before the frame."""
    import torch
    from r2s_hip import synth

    dev = torch.device("cuda:0")
    W, H = 640, 480
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    cams = [synth.side_camera(W, H), synth.wrist_camera(W, H)]
    scenes = [synth.gaussian_scene(7000, 21), synth.gaussian_scene(4000, 22)]
    frame_of = [(0, 0, None), (1, 1, None), (1, 0, None), (0, 1, None)]
    n, pl, rg, _, _ = _render(_ctx(dev, False), scenes, cams, frame_of, W, H, dev, False)
    base_inst, base_g = 0, 0
    rg = rg.reshape(len(frame_of), tiles, 2)
    for f, (si, ci, _) in enumerate(frame_of):
        n_ref, _, _, _, dbg = oracle_render(scenes[si], cams[ci], debug=True)
        exp_r = dbg["ranges"].astype(np.int64)
        assert np.array_equal(rg[f], np.where((exp_r[:, 1] > exp_r[:, 0])[:, None], exp_r + base_inst, 0)), f
        assert np.array_equal(pl[base_inst:base_inst + n_ref], dbg["point_list"].astype(np.int64) + base_g), f
        base_inst += n_ref
        base_g += scenes[si]["means3D"].shape[0]
    assert base_inst == n


def test_more_than_2048_tiles_keeps_the_radix_sort_and_the_same_pixels():
    """1280 x 720 = 3 600 tiles per frame: beyond the LDS counters of the binning, the library takes the radix sort by itself."""
    import torch
    from r2s_hip import synth

    dev = torch.device("cuda:0")
    W, H = 1280, 720
    cams = [synth.side_camera(W, H)]
    scenes = [synth.gaussian_scene(6000, 31)]
    a = _render(_ctx(dev, False), scenes, cams, [(0, 0, None)], W, H, dev, True)
    b = _render(_ctx(dev, True), scenes, cams, [(0, 0, None)], W, H, dev, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


def test_sync_free_batches_bin_with_a_capacity_and_report_an_overflow():
    """Sync-free mode: the second batch bins over the capacity derived from the first (chunk table built on the device, workgroups past the
    last chunk leave) and produces the same pixels; a batch that outgrows the capacity raises the overflow word, and the call after it
    sizes itself again."""
    import torch
    from r2s_hip import synth

    dev = torch.device("cuda:0")
    W, H = 640, 480
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    cam = synth.side_camera(W, H)
    small, big = synth.gaussian_scene(4000, 41), synth.gaussian_scene(16000, 42)
    rb = _ctx(dev, False)
    rb.set_tile_culling(True)

    def frames_for(rb, sc, out_c, out_d):
        s = rb.make_set(t(sc["means3D"]), t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
        fr = [dict(set=0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), bg=t(cam["bg"]),
                   tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], z_threshold=cam["z_threshold"], out_color=out_c[k], out_depth=out_d[k]) for k in range(2)]
        return [s], fr

    ref_c = torch.empty(2, 3, H, W, device=dev); ref_d = torch.empty(2, 1, H, W, device=dev)
    n_small = rb.forward(*frames_for(rb, small, ref_c, ref_d), W, H)
    torch.cuda.synchronize()
    rb.set_async(True)
    out_c = torch.empty(2, 3, H, W, device=dev); out_d = torch.empty(2, 1, H, W, device=dev)
    args = frames_for(rb, small, out_c, out_d)
    rb.forward(*args, W, H)        # first call of the mode: reads the count once, fixes the capacity
    out_c.zero_()
    rb.forward(*args, W, H)        # sync-free over the capacity
    rc, n, over = rb.poll(wait=True)
    assert rc == 0 and n == n_small and over == 0
    assert torch.equal(out_c, ref_c) and torch.equal(out_d, ref_d)
    big_c = torch.empty(2, 3, H, W, device=dev); big_d = torch.empty(2, 1, H, W, device=dev)
    big_args = frames_for(rb, big, big_c, big_d)
    rb.forward(*big_args, W, H)    # four times the instances: beyond the capacity
    rc, n_big, over = rb.poll(wait=True)       # an overflowing batch reports the emission slots it needed (>= its instance count)
    assert over == 1 and n_big > n_small * 2
    rb.forward(*big_args, W, H)    # re-sized by synchronising once
    rc, n2, over2 = rb.poll(wait=True)
    exp_c = torch.empty(2, 3, H, W, device=dev); exp_d = torch.empty(2, 1, H, W, device=dev)
    rb2 = _ctx(dev, True); rb2.set_tile_culling(True)
    n_exact = rb2.forward(*frames_for(rb2, big, exp_c, exp_d), W, H)   # radix-sort path: the survivors are counted before the emission
    torch.cuda.synchronize()
    assert over2 == 1 and n2 == n_exact and n_big >= n_exact
    assert torch.equal(big_c, exp_c) and torch.equal(big_d, exp_d)


def test_binning_with_poisoned_allocations_in_a_fresh_process():
    """R2S_POISON=1 (every device allocation of the library filled with 0xFF, read once per process): the binning must not read a histogram
    row, chunk descriptor or class counter it never wrote — the batch comparison above, run again in a poisoned subprocess."""
    import subprocess
    import sys

    here = os.path.abspath(__file__)
    env = dict(os.environ, R2S_POISON="1")
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-k", "equals_radix_sort_path and 640", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0 and "2 passed" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-800:])


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_binning_equals_radix_sort_path_on_random_shapes(seed):
    """Odd shapes: widths / heights that are no multiple of the tile, one to seven frames, sets of 1 to a few thousand Gaussians (frames far
    below one chunk, sets of different sizes in one batch), culling on and off — both paths, identical counts, lists, ranges and pixels."""
    import torch
    from r2s_hip import synth

    rng = np.random.default_rng(1000 + seed)
    dev = torch.device("cuda:0")
    W, H = int(rng.integers(40, 700)), int(rng.integers(30, 500))
    cams = [synth.side_camera(W, H), synth.wrist_camera(W, H)]
    sizes = [int(rng.choice([1, 7, 64, 300, 2500, 6000])) for _ in range(int(rng.integers(1, 4)))]
    scenes = [synth.gaussian_scene(max(p, 3), 200 + 10 * seed + k) for k, p in enumerate(sizes)]
    for sc, p in zip(scenes, sizes):      # cut to the drawn size (gaussian_scene needs a few points to lay out its table)
        for key in ("means3D", "opacities", "shs", "scales", "rotations"):
            sc[key] = np.ascontiguousarray(sc[key][:p])
    frame_of = [(int(rng.integers(0, len(scenes))), int(rng.integers(0, 2)), None) for _ in range(int(rng.integers(1, 8)))]
    for cull in (False, True):
        a = _render(_ctx(dev, False), scenes, cams, frame_of, W, H, dev, cull)
        b = _render(_ctx(dev, True), scenes, cams, frame_of, W, H, dev, cull)
        assert a[0] == b[0], (W, H, sizes, frame_of, cull)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (W, H, sizes, frame_of, cull)
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), (W, H, sizes, frame_of, cull)
