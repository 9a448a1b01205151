"""Row f3 on the device: a scene assembled from FILES — object splat PLY, table + robot scan PLY with its link-mask .npy, a
static mesh with its own splat (GSRenderer.load_scaniverse, gs_renderer.py:333-714, via GSProcessor.load) — concatenated like
update_rendervar (gs_renderer.py:886-917), rendered on the HIP path and compared with the oracle; the link mask drives the
robot-Gaussian placement."""
import numpy as np
import pytest

from test_assets import scaniverse_scene
from util_raster import compare_images, oracle_render

pytestmark = pytest.mark.gpu


def test_scene_loaded_from_ply_files_renders_like_the_oracle(tmp_path):
    import torch
    from r2s_hip import assets, synth
    from r2s_hip.raster import RasterBatch
    from r2s_hip.robot import RobotGaussians
    from sim.utils.gs.gs_processor import GSProcessor

    cfg, src = scaniverse_scene(tmp_path, n_obj=3000, n_tab=5000, n_box=800)
    # the drop-in GSProcessor reads the same files the reference's does (gs_processor.py:59-100)
    p = GSProcessor().load(cfg["object"]["path"])
    assert tuple(p["means3D"].shape) == (3000, 3) and tuple(p["sh_colors"].shape) == (3000, 48)
    sc = assets.load_scaniverse(cfg, randomize=True, index=7)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    # update_rendervar: object, then the static meshes' splats, then the table / robot scan; every rotation normalised (:906)
    # (assets.assemble_rendervar: pinned by the reference's own update_rendervar, tests/test_assets.py / tests/golden/scene_assembly.npz)
    cat = assets.assemble_rendervar(sc["rendervar"], sc["params_meshes"], sc["table_rendervar"])
    n_front = len(cat["means3D"]) - len(sc["table_rendervar"]["means3D"])
    means = t(cat["means3D"]).clone()
    rots = t(cat["rotations"]).clone()
    # the robot part of the scan follows its links: mask from the .npy, link poses from a stand-in FK
    mask = sc["total_mask_full"].astype(np.int32)
    n_links = 11
    rng = np.random.default_rng(5)
    offsets = np.stack([np.eye(4)] * n_links)
    base = np.stack([np.eye(4, dtype=np.float32)] * n_links)
    base[:, :3, 3] = rng.uniform(-0.2, 0.2, (n_links, 3))
    pose = base.copy(); pose[:, :3, 3] += np.array([0.0, 0.0, 0.03], np.float32)         # every link 3 cm up
    rg = RobotGaussians(n_links, (1, 2, 3, 4, 5, 6, 7, 8), offsets, base, sc["table_rendervar"]["means3D"], sc["table_rendervar"]["rotations"], mask, device=dev)
    rg.transform(t(pose)[None], means[None, n_front:], rots[None, n_front:], normalize=True, write_static=True)
    torch.cuda.synchronize()
    moved = means[n_front:].cpu().numpy() - sc["table_rendervar"]["means3D"]
    on_link = np.isin(mask, (1, 2, 3, 4, 5, 6, 7, 8))
    assert np.allclose(moved[on_link], [0.0, 0.0, 0.03], atol=1e-6) and np.abs(moved[~on_link]).max() == 0.0
    # render: use_shs False -> DC band only (gs_renderer.py:944-947)
    W, H = 320, 240
    cam = synth.side_camera(W, H)
    rb = RasterBatch(dev)
    g = dict(opacities=t(cat["opacities"]), shs=t(cat["shs"][:, 0:1]), scales=t(cat["scales"]))
    s = rb.make_set(means, g["opacities"], shs=g["shs"], scales=g["scales"], rotations=rots)
    out_c = torch.empty(3, H, W, device=dev); out_d = torch.empty(1, H, W, device=dev)
    fr = dict(set=0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), bg=t(cam["bg"]), tanfovx=cam["tanfovx"],
              tanfovy=cam["tanfovy"], z_threshold=cam["z_threshold"], out_color=out_c, out_depth=out_d)
    n = rb.forward([s], [fr], W, H)
    torch.cuda.synchronize()
    scene = dict(means3D=means.cpu().numpy(), opacities=cat["opacities"], shs=cat["shs"][:, 0:1], scales=cat["scales"], rotations=rots.cpu().numpy())
    n_ref, col_ref, _, dep_ref = oracle_render(scene, cam)
    assert n == n_ref and n > 0
    r = compare_images(out_c.cpu().numpy(), out_d.cpu().numpy(), col_ref, dep_ref, what="scene assembled from PLY files + link mask, vs oracle")
    assert r["frac_rgb"] <= 1e-4 and r["frac_depth"] <= 1e-4, r
    assert col_ref.std() > 0
