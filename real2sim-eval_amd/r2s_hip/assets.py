"""Asset formats either side of the hot paths: the INRIA / Scaniverse Gaussian-splat PLY that
``GSProcessor.load / save`` read and write (sim/utils/gs/gs_processor.py:59-171) and the PhysTwin case directory that
``SpringMassDynamicsModule.__init__`` loads (sim/physics/phystwin.py:231-298).  Host code; the PLY codec is a
self-contained numpy reader / writer of the binary_little_endian (and ascii) vertex element — the reference uses the
``plyfile`` package, which this image does not have."""
from __future__ import annotations

import glob
import os
import pickle as pkl

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}

GS_FIELDS = (["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + ["opacity", "scale_0", "scale_1", "scale_2",
             "rot_0", "rot_1", "rot_2", "rot_3"])   # property order GSProcessor.save writes (gs_processor.py:157-167)


def read_ply_vertices(path) -> np.ndarray:
    """The ``vertex`` element of a PLY file as a numpy structured array (scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = dict(name=tok[1], count=int(tok[2]), props=[])
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    cur["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("binary_little_endian", "binary_big_endian", "ascii"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        for el in elements:
            if any(p[0] == "list" for p in el["props"]):
                if el["name"] == "vertex":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                break  # a list element before the vertices would need a walk; splat files put the vertices first
            order = "<" if fmt != "binary_big_endian" else ">"
            dt = np.dtype([(n, order + t) for n, t in el["props"]])
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(el["count"])]
                arr = np.zeros(el["count"], dtype=dt)
                for k, (n, _) in enumerate(el["props"]):
                    arr[n] = np.array([r[k] for r in rows], dtype=np.float64).astype(dt[n])
            else:
                arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
            if el["name"] == "vertex":
                return arr
    raise ValueError(f"{path}: no vertex element")


def write_ply_vertices(path, vertex: np.ndarray):
    """binary_little_endian PLY with one ``vertex`` element, header laid out like plyfile writes it."""
    names = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint"}
    v = np.ascontiguousarray(vertex.astype(vertex.dtype.newbyteorder("<")))
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {len(v)}"]
    head += [f"property {names[v.dtype[n].str[1:]]} {n}" for n in v.dtype.names]
    head.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(v.tobytes())


def load_gaussians_ply(path, rot_x_minus90=False):
    """``GSProcessor.load`` (gs_processor.py:59-100): dict of numpy float32 arrays means3D [n,3], sh_colors [n,48],
    log_scales [n,3], unnorm_rotations [n,4] (w,x,y,z), logit_opacities [n,1]."""
    v = read_ply_vertices(path)
    col = lambda n: np.asarray(v[n], np.float32)  # noqa: E731
    pts = np.stack([col("x"), col("y"), col("z")], -1)
    sh = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")] + [col(f"f_rest_{i}") for i in range(45)], -1)
    quats = np.stack([col(f"rot_{i}") for i in range(4)], -1)
    if rot_x_minus90:  # make z the up axis (:88-92)
        Rm = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
        pts = np.dot(Rm, pts.T).T.astype(np.float32)
        q0 = rot_mat_to_quat(Rm)
        quats = quat_mult(q0, quats) if len(quats) else quats
    return dict(means3D=pts, sh_colors=sh, log_scales=np.stack([col(f"scale_{i}") for i in range(3)], -1), unnorm_rotations=quats,
                logit_opacities=col("opacity")[:, None])


def save_gaussians_ply(params, path):
    """``GSProcessor.save`` (gs_processor.py:139-171)."""
    n = len(params["means3D"])
    colors = np.asarray(params["sh_colors"], np.float32).reshape(n, -1)
    fields = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(colors.shape[1] - 3)] + GS_FIELDS[-8:]
    out = np.zeros(n, dtype=[(f, "<f4") for f in fields])
    data = np.concatenate([np.asarray(params["means3D"], np.float32), colors, np.asarray(params["logit_opacities"], np.float32).reshape(n, 1),
                           np.asarray(params["log_scales"], np.float32), np.asarray(params["unnorm_rotations"], np.float32)], axis=1)
    for k, f in enumerate(fields):
        out[f] = data[:, k]
    write_ply_vertices(path, out)


def quat_mult(q1, q2):
    """Hamilton product, (w, x, y, z); broadcasts over leading axes (robot_pc_sampler.py quat_mult, kornia convention)."""
    q1, q2 = np.asarray(q1, np.float32), np.asarray(q2, np.float32)
    w1, x1, y1, z1 = q1[..., 0], q1[..., 1], q1[..., 2], q1[..., 3]
    w2, x2, y2, z2 = q2[..., 0], q2[..., 1], q2[..., 2], q2[..., 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1).astype(np.float32)


def rot_mats_to_quats(R):
    """Rotation matrices [..., 3, 3] -> unit quaternions [..., 4] (w, x, y, z), the four-branch form selected per matrix
    (trace > 0, else the largest diagonal element), vectorised: a scene is 1e5-1e6 splats."""
    R = np.asarray(R, np.float64)
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    t = m00 + m11 + m22
    with np.errstate(invalid="ignore", divide="ignore"):
        s0 = np.sqrt(t + 1.0) * 2
        q0 = np.stack([0.25 * s0, (R[..., 2, 1] - R[..., 1, 2]) / s0, (R[..., 0, 2] - R[..., 2, 0]) / s0, (R[..., 1, 0] - R[..., 0, 1]) / s0], -1)
        s1 = np.sqrt(1.0 + m00 - m11 - m22) * 2
        q1 = np.stack([(R[..., 2, 1] - R[..., 1, 2]) / s1, 0.25 * s1, (R[..., 0, 1] + R[..., 1, 0]) / s1, (R[..., 0, 2] + R[..., 2, 0]) / s1], -1)
        s2 = np.sqrt(1.0 + m11 - m00 - m22) * 2
        q2 = np.stack([(R[..., 0, 2] - R[..., 2, 0]) / s2, (R[..., 0, 1] + R[..., 1, 0]) / s2, 0.25 * s2, (R[..., 1, 2] + R[..., 2, 1]) / s2], -1)
        s3 = np.sqrt(1.0 + m22 - m00 - m11) * 2
        q3 = np.stack([(R[..., 1, 0] - R[..., 0, 1]) / s3, (R[..., 0, 2] + R[..., 2, 0]) / s3, (R[..., 1, 2] + R[..., 2, 1]) / s3, 0.25 * s3], -1)
    c0 = t > 0
    c1 = ~c0 & (m00 > m11) & (m00 > m22)
    c2 = ~c0 & ~c1 & (m11 > m22)
    q = np.where(c0[..., None], q0, np.where(c1[..., None], q1, np.where(c2[..., None], q2, q3)))
    return q.astype(np.float32)


def rot_mat_to_quat(R):
    """One rotation matrix -> unit quaternion (w, x, y, z)."""
    return rot_mats_to_quats(np.asarray(R, np.float64)[None])[0]


def render_inputs_from_params(params, use_shs=False):
    """What GSRenderer hands to the rasteriser from loaded params (gs_renderer.py:897-917): normalised rotations,
    exp(log_scales), sigmoid(logit_opacities), shs [n,1,3] (DC only) or [n,16,3]."""
    q = np.asarray(params["unnorm_rotations"], np.float32)
    q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-12)
    sh = np.asarray(params["sh_colors"], np.float32)
    shs = (np.concatenate([sh[:, :3][:, None], sh[:, 3:].reshape(len(sh), 3, -1).transpose(0, 2, 1)], axis=1) if use_shs else sh[:, None, :3])
    return dict(means3D=np.asarray(params["means3D"], np.float32), rotations=q.astype(np.float32), scales=np.exp(np.asarray(params["log_scales"], np.float32)),
                opacities=(1.0 / (1.0 + np.exp(-np.asarray(params["logit_opacities"], np.float32)))).astype(np.float32), shs=np.ascontiguousarray(shs, np.float32))


# ---- PhysTwin case directory ------------------------------------------------------------------------------------------
_RENAMES = {"global_spring_Y": "init_spring_Y", "collide_object_elas": "collide_self_elas", "collide_object_fric": "collide_self_fric"}


def load_phystwin_case(data_path, zeroth_order_ckpt_path, first_order_ckpt_path, case_name, init_pose=None, object_radius=0.02,
                       object_max_neighbours=30):
    """What SpringMassDynamicsModule.__init__ reads (phystwin.py:231-298):
      {data_path}/{case}/final_data.pkl              object_points [T,N,3], surface_points, interior_points
      {zeroth}/{case}/optimal_params.pkl             scalar physics parameters (renamed like :251-255)
      {first}/{case}/train/best_*.pth                spring_Y (log stiffness), collide_*, num_object_springs
    Returns dict(points float32 [n,3] (aligned by init_pose), springs int32 [S,2], rest float32 [S], spring_Y float32 [S],
    collide_elas/fric, collide_self_elas/fric (float), params dict).  Springs are rebuilt with the reference's
    procedure (hybrid radius / k-nearest search, de-duplicated, rest > 1e-4) and must number num_object_springs."""
    import torch

    from .synth import build_springs

    with open(os.path.join(data_path, case_name, "final_data.pkl"), "rb") as f:
        data = pkl.load(f)
    object_pts = np.concatenate([np.asarray(data["object_points"])[0], np.asarray(data["surface_points"]), np.asarray(data["interior_points"])], axis=0)
    pose = np.eye(4) if init_pose is None else np.asarray(init_pose, np.float64)
    aligned = object_pts @ pose[:3, :3].T + pose[:3, 3]
    with open(os.path.join(zeroth_order_ckpt_path, case_name, "optimal_params.pkl"), "rb") as f:
        optimal = dict(pkl.load(f))
    params = {_RENAMES.get(k, k): (v.item() if hasattr(v, "item") else v) for k, v in optimal.items()}
    springs, _ = build_springs(object_pts, object_radius, object_max_neighbours)
    keep = np.linalg.norm(aligned[springs[:, 0]] - aligned[springs[:, 1]], axis=1) > 1e-4
    springs = springs[keep]
    p32 = aligned.astype(np.float32)
    rest = np.linalg.norm(p32[springs[:, 0]] - p32[springs[:, 1]], axis=1).astype(np.float32)
    best = sorted(glob.glob(os.path.join(first_order_ckpt_path, case_name, "train", "best_*.pth")))
    if not best:
        raise FileNotFoundError(f"no best_*.pth under {first_order_ckpt_path}/{case_name}/train")
    ck = torch.load(best[0], map_location="cpu", weights_only=False)
    n_obj = int(ck["num_object_springs"])
    if len(springs) != n_obj:
        raise ValueError(f"{case_name}: rebuilt {len(springs)} springs, checkpoint has {n_obj} object springs")
    sc = lambda k: float(np.asarray(ck[k].detach().cpu() if hasattr(ck[k], "detach") else ck[k]).reshape(-1)[0])  # noqa: E731
    sy = ck["spring_Y"]
    sy = (sy.detach().cpu().numpy() if hasattr(sy, "detach") else np.asarray(sy)).astype(np.float32)[:n_obj]
    return dict(points=p32, springs=springs.astype(np.int32), rest=rest, spring_Y=sy, collide_elas=sc("collide_elas"), collide_fric=sc("collide_fric"),
                collide_self_elas=sc("collide_object_elas"), collide_self_fric=sc("collide_object_fric"), params=params)


# ---- scene colour correction of the SH coefficients (gs_renderer.py:656-699) -----------------------------------------------
C0 = 0.28209479177387814


def sh_colors_to_shs(sh_colors):
    """(n, 48) file layout (3 DC, then 45 rest grouped per channel) -> (n, 16, 3) coefficient-major (gs_renderer.py:650-654)."""
    sh = np.asarray(sh_colors)
    n = sh.shape[0]
    return np.concatenate([sh[:, :3][:, None], sh[:, 3:].reshape(n, 3, -1).transpose((0, 2, 1))], axis=1)


def color_correct_shs(shs, color_A, color_b):
    """The scene's colour calibration applied to SH coefficients so that the RENDERED colour c = C0 * sh0 + 0.5 becomes
    A c + b (``color_A`` 3x3) or A2 c^2 + A1 c + b (``color_A`` 3x6 = [A2 | A1], DC band only for the square term), bands
    >= 1 transformed by the linear part.  shs: (n, K, 3) with K = (deg + 1)^2."""
    shs = np.asarray(shs)
    A = np.array(color_A, dtype=np.float32).reshape(3, -1)
    b = np.array(color_b, dtype=np.float32).reshape(3)
    deg = int(np.sqrt(shs.shape[1]) - 1)
    out = []
    if A.shape[1] == 3:
        for si in range(deg + 1):
            band = shs[:, si ** 2:(si + 1) ** 2, :]
            if si == 0:
                off = np.ones(3) * 0.5
                bias = (1.0 / C0) * (off.reshape(1, 3) @ A.T + b - off)
                out.append((np.squeeze(band, axis=1) @ A.T + bias)[:, None])
            else:
                out.append(band @ A.T)
    elif A.shape[1] == 6:
        A_2, A_1 = A[:, :3], A[:, 3:]
        for si in range(deg + 1):
            band = shs[:, si ** 2:(si + 1) ** 2, :]
            if si == 0:
                flat = np.squeeze(band, axis=1)
                o1, o2 = np.ones(3) * 0.5, np.ones(3) * 0.25
                bias = (1.0 / C0) * (o2.reshape(1, 3) @ A_2.T + o1.reshape(1, 3) @ A_1.T + b - o1)
                out.append((flat @ A_1.T + (flat + C0 * flat ** 2) @ A_2.T + bias)[:, None])
            else:
                out.append(band @ A_1.T)
    else:
        raise ValueError("color_A must have 9 (linear) or 18 (quadratic) entries")
    return np.concatenate(out, axis=1)


# ---- scene assembly from files: GSRenderer.load_scaniverse (gs_renderer.py:333-714) ----------------------------------------
def read_triangle_mesh(path):
    """(vertices float64 [nv,3], triangles int32 [nf,3]) of a binary / ascii STL or a Wavefront OBJ — the role of
    ``o3d.io.read_triangle_mesh`` at gs_renderer.py:358 (open3d is not a dependency here).  STL corners are kept as stored
    (one vertex per corner: the stepper welds coincident vertices itself, r2s_phys_create)."""
    path = str(path)
    if path.lower().endswith(".obj"):
        v, f = [], []
        with open(path) as fh:
            for line in fh:
                t = line.split()
                if not t:
                    continue
                if t[0] == "v":
                    v.append([float(x) for x in t[1:4]])
                elif t[0] == "f":
                    idx = [int(x.split("/")[0]) for x in t[1:]]
                    idx = [i - 1 if i > 0 else len(v) + i for i in idx]
                    for k in range(1, len(idx) - 1):   # fan-triangulate polygons
                        f.append([idx[0], idx[k], idx[k + 1]])
        return np.asarray(v, np.float64).reshape(-1, 3), np.asarray(f, np.int32).reshape(-1, 3)
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[80:84], "<u4")[0]) if len(raw) >= 84 else -1
    if n >= 0 and len(raw) == 84 + 50 * n:            # binary STL: 80-byte header, count, 50-byte records
        rec = np.frombuffer(raw[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n)
        return rec["v"].reshape(-1, 3).astype(np.float64), np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    v = [[float(x) for x in line.split()[1:4]] for line in raw.decode("ascii", "replace").splitlines() if line.strip().startswith("vertex")]
    return np.asarray(v, np.float64).reshape(-1, 3), np.arange(len(v), dtype=np.int32).reshape(-1, 3)


def quats_to_rot_mats(q):
    """Unit quaternions (w, x, y, z) [..., 4] -> rotation matrices (kornia's quaternion_to_rotation_matrix, which normalises first)."""
    q = np.asarray(q, np.float64)
    q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-12)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def _grid_pose(grid, idx):
    """One entry of a grid randomisation (cfg.gs.*.grid_randomization: xy list, theta list in degrees, one_to_one):
    (x, y, z = 0, angle in radians), gs_renderer.py:376-388 / :626-637."""
    xy, theta = grid["xy"], grid["theta"]
    if grid.get("one_to_one", False):
        return float(xy[idx][0]), float(xy[idx][1]), 0.0, float(theta[idx]) * np.pi / 180.0
    xi, ti = idx // len(theta), idx % len(theta)
    return float(xy[xi][0]), float(xy[xi][1]), 0.0, float(theta[ti]) * np.pi / 180.0


def _randomised_pose(pose, entry, randomize, use_grid, grid_index, rng, random_variables, is_mesh=False):
    """The reference's two condition chains differ: a static mesh without a grid of its own is STILL drawn from its
    translation_range / azimuth_range under use_grid_randomization (`elif randomize:`, gs_renderer.py:393), the object only when
    grid randomisation is off (`elif randomize and not use_grid_randomization`, :639)."""
    pose = np.array(pose, dtype=np.float64).reshape(4, 4).copy()
    rand = None
    if randomize and use_grid and entry.get("grid_randomization"):
        rand = _grid_pose(entry["grid_randomization"], grid_index)
    elif randomize and (is_mesh or not use_grid):
        tr, az = np.array(entry["translation_range"]), np.array(entry["azimuth_range"])
        rand = (rng.uniform(tr[0], tr[1]), rng.uniform(tr[2], tr[3]), rng.uniform(tr[4], tr[5]), rng.uniform(az[0], az[1]) * np.pi / 180.0)
    if rand is not None:
        x, y, z, a = rand
        pose[:3, 3] += np.array([x, y, z], dtype=np.float32)
        rot_z = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
        pose[:3, :3] = rot_z @ pose[:3, :3]
        random_variables.append([x, y, z, a])
    return pose


def _splat_arrays(params, entry):
    """means, shs [n,16,3] (colour-corrected when the entry has color_A / color_b), scales, raw quaternions, opacities of a
    loaded splat file (gs_renderer.py:417-462, :543-590, :639-699)."""
    shs = sh_colors_to_shs(np.asarray(params["sh_colors"], np.float32))
    if "color_A" in entry:
        shs = color_correct_shs(shs, entry["color_A"], entry["color_b"])
    return (np.asarray(params["means3D"], np.float32), shs.astype(np.float32), np.exp(np.asarray(params["log_scales"], np.float32)),
            np.asarray(params["unnorm_rotations"], np.float32), (1.0 / (1.0 + np.exp(-np.asarray(params["logit_opacities"], np.float32)))).astype(np.float32))


def load_scaniverse(gs_cfg, randomize=False, index=None, rng=None):
    """``GSRenderer.load_scaniverse`` (gs_renderer.py:333-714) without the renderer object: assemble the scene of one episode from
    its files.  ``gs_cfg`` mirrors ``cfg.gs``: {'object': {'path', 'pose', ['color_A', 'color_b'], ['grid_randomization' |
    'translation_range', 'azimuth_range']}, 'scene': {'table_splat_path', 'total_mask_path', ['color_A', 'color_b']}, 'meshes':
    [{'name', 'mesh_path', 'splat_path', 'pose', ...}], 'use_grid_randomization': bool}.  Returns numpy arrays:
      rendervar        object splats in the world frame: means3D posed by the (randomised) object pose, rotations =
                       normalise(quat(pose_R . R(q))), scales, opacities, shs [n,16,3]                              (:592-648)
      table_rendervar  table + robot scan as stored (rotations NOT normalised, like the reference keeps them)        (:650-714)
      params_meshes    per static mesh: its splats (positions posed, rotations only normalised), and
      meshes           its collision mesh (vertices posed, triangles)                                                (:354-501)
      total_mask_full  link id of every table / robot splat (float32, the reference's dtype)                          (:503-505)
      pose_obj, random_variables
    The episode ``index`` is decoded like the reference: with grid randomisation the object takes index % n_object_rand and the
    meshes share index // n_object_rand, peeled mesh by mesh (:343-352, :368-371).  ``rng``: uniform randomisation draws from it in the
    reference's order (meshes first, then the object; x, y, z, angle each) — ``np.random.RandomState(seed)`` reproduces the reference's
    ``np.random.seed(seed)`` (env.py:32).  Pinned: tests/golden/scene_assembly.npz holds what the reference's own function returned for
    a scene directory (tests/golden/make_scene_golden.py), tests/test_assets.py compares."""
    rng = np.random.default_rng() if rng is None else rng
    use_grid = bool(gs_cfg.get("use_grid_randomization", False))
    obj_cfg, scene_cfg = gs_cfg["object"], gs_cfg["scene"]
    params_obj = load_gaussians_ply(obj_cfg["path"])
    params_table = load_gaussians_ply(scene_cfg["table_splat_path"])
    true_index, true_index_mesh = index, None
    if randomize and use_grid:
        g = obj_cfg["grid_randomization"]
        n_obj_rand = len(g["xy"]) if g.get("one_to_one", False) else len(g["xy"]) * len(g["theta"])
        assert index is not None
        true_index_mesh, true_index = index // n_obj_rand, index % n_obj_rand
    random_variables, params_meshes, meshes = [], {}, {}
    for m in gs_cfg.get("meshes", []):
        gi = None
        if randomize and use_grid and m.get("grid_randomization"):
            g = m["grid_randomization"]
            n_this = len(g["xy"]) if g.get("one_to_one", False) else len(g["xy"]) * len(g["theta"])
            gi, true_index_mesh = true_index_mesh % n_this, true_index_mesh // n_this
        pose = _randomised_pose(m["pose"], m, randomize, use_grid, gi, rng, random_variables, is_mesh=True)
        pts, shs, scales, quats, opac = _splat_arrays(load_gaussians_ply(m["splat_path"]), m)
        pts = pts @ pose[:3, :3].T + pose[:3, 3]
        q = quats / np.maximum(np.linalg.norm(quats, axis=-1, keepdims=True), 1e-12)
        params_meshes[m["name"]] = dict(means3D=pts.astype(np.float32), shs=shs, scales=scales, rotations=q.astype(np.float32), opacities=opac)
        v, f = read_triangle_mesh(m["mesh_path"])
        meshes[m["name"]] = ((v @ pose[:3, :3].T + pose[:3, 3]).astype(np.float32), f)
    total_mask_full = np.load(scene_cfg["total_mask_path"]).astype(np.float32)
    # object: posed by cfg.gs.object.pose (+ randomisation), rotations composed with the pose's rotation
    pose_obj = _randomised_pose(obj_cfg["pose"], obj_cfg, randomize, use_grid, true_index, rng, random_variables).astype(np.float32)
    pts, shs, scales, quats, opac = _splat_arrays(params_obj, obj_cfg)
    qn = quats / np.maximum(np.linalg.norm(quats, axis=-1, keepdims=True), 1e-12)
    rot = pose_obj[:3, :3].astype(np.float64) @ quats_to_rot_mats(qn)
    q_world = rot_mats_to_quats(rot)
    q_world = q_world / np.maximum(np.linalg.norm(q_world, axis=-1, keepdims=True), 1e-12)
    rendervar = dict(means3D=(pts @ pose_obj[:3, :3].T + pose_obj[:3, 3]).astype(np.float32), shs=shs, scales=scales, rotations=q_world.astype(np.float32),
                     opacities=opac)
    tpts, tshs, tscales, tquats, topac = _splat_arrays(params_table, scene_cfg)
    if len(total_mask_full) != len(tpts):
        raise ValueError(f"total_mask has {len(total_mask_full)} entries, the table / robot scan {len(tpts)} splats")
    table_rendervar = dict(means3D=tpts, shs=tshs, scales=tscales, rotations=tquats, opacities=topac)
    return dict(rendervar=rendervar, table_rendervar=table_rendervar, params_meshes=params_meshes, meshes=meshes, total_mask_full=total_mask_full,
                pose_obj=pose_obj, random_variables=random_variables)


def assemble_rendervar(rendervar, params_meshes, table_params):
    """``GSRenderer.update_rendervar``'s scene assembly (gs_renderer.py:758-769, :797-813, :886-921): the object's splats (already skinned to
    the particles' new positions; their rotations normalised, :758), then every static mesh's splats in the order of ``params_meshes``,
    then the table + robot scan as ``transform_gs_xarm_gripper / _pusher`` returned it; every rotation of the result normalised
    (:906).  Numpy arrays in, the ``rendervar_full`` dictionary out (without the unused ``means2D``)."""
    parts = [dict(rendervar)] + [params_meshes[k] for k in params_meshes] + [table_params]
    out = {k: np.concatenate([np.asarray(q[k], np.float32) for q in parts]) for k in ("means3D", "shs", "rotations", "opacities", "scales")}
    r = out["rotations"].astype(np.float32)
    out["rotations"] = r / np.maximum(np.linalg.norm(r, axis=-1, keepdims=True), 1e-12).astype(np.float32)
    return out
