"""Host-side pieces added in round 3 that need no GPU: the rotation helpers of the action path (phystwin.py:131-138, :377-380),
the source hash that keys the counter summaries, and bench.py's stale-summary labelling."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rotation_helpers_of_the_action_path_against_scipy():
    import torch
    from scipy.spatial.transform import Rotation
    from r2s_hip.rollout import axis_angle_to_rotation_matrix, rotation_matrix_to_axis_angle

    rng = np.random.default_rng(0)
    aa = rng.normal(0, 0.8, (64, 3)).astype(np.float32)
    aa[0] = 0.0                      # first-order branch (theta^2 <= 1e-6)
    aa[1] = (3e-4, -2e-4, 1e-4)
    R = axis_angle_to_rotation_matrix(torch.from_numpy(aa)).numpy()
    assert np.abs(R - Rotation.from_rotvec(aa.astype(np.float64)).as_matrix()).max() < 2e-6
    back = rotation_matrix_to_axis_angle(torch.from_numpy(R)).numpy()
    assert np.abs(back - aa).max() < 5e-6
    # phystwin.py:136-137: the rate that takes eef_rot to eef_rot_next is the log of eef_rot . inv(eef_rot_next)
    R0 = Rotation.from_rotvec(rng.normal(0, 1.0, (8, 3))).as_matrix()
    step = rng.normal(0, 0.05, (8, 3))
    R1 = np.einsum("nij,njk->nik", Rotation.from_rotvec(step).as_matrix().transpose(0, 2, 1), R0)     # :380: next = delta^T . rot
    got = rotation_matrix_to_axis_angle(torch.from_numpy(np.einsum("nij,njk->nik", R0, np.linalg.inv(R1)).astype(np.float32))).numpy()
    assert np.abs(got - step).max() < 2e-6


def test_rotation_matrix_to_quaternion_of_the_robot_state_against_scipy():
    """obs['robot']['eef_quat'] (env.py:62-66): scalar-first quaternion of the end-effector rotation.  All four branches of the
    conversion (trace > 0; each diagonal entry the largest), against scipy up to the sign, and back to the matrix."""
    import torch
    from scipy.spatial.transform import Rotation
    from r2s_hip.rollout import quaternion_to_rotation_matrix, rotation_matrix_to_quaternion

    rng = np.random.default_rng(2)
    rv = np.concatenate([rng.normal(0, 0.5, (32, 3)),                                     # trace > 0
                         np.pi * 0.98 * np.eye(3), np.pi * 0.9 * np.eye(3)[[0, 1, 2, 0]] + 0.05,  # half turns: one branch per axis
                         rng.normal(0, 2.5, (64, 3)), np.zeros((1, 3))])
    Rm = Rotation.from_rotvec(rv).as_matrix()
    q = rotation_matrix_to_quaternion(torch.from_numpy(Rm.astype(np.float32))).numpy().astype(np.float64)
    assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 1e-6
    ref = Rotation.from_matrix(Rm).as_quat()[:, [3, 0, 1, 2]]                                # scipy: x, y, z, w
    sign = np.sign((q * ref).sum(1, keepdims=True))
    assert np.abs(q - sign * ref).max() < 2e-6
    assert np.abs(Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix() - Rm).max() < 2e-6
    back = quaternion_to_rotation_matrix(torch.from_numpy((q * 3.0).astype(np.float32))).numpy()      # normalises its input
    assert np.abs(back - Rm).max() < 2e-6
    tr = np.trace(Rm, axis1=1, axis2=2)
    used = {"trace": int((tr > 0).sum())}
    d = np.stack([Rm[:, 0, 0], Rm[:, 1, 1], Rm[:, 2, 2]], 1)
    for k in range(3):
        used[f"pivot {k}"] = int(((tr <= 0) & (d.argmax(1) == k)).sum())
    assert all(v > 0 for v in used.values()), used
    assert np.array_equal(q[-1], [1.0, 0.0, 0.0, 0.0]) or np.abs(q[-1] - [1, 0, 0, 0]).max() < 1e-7   # identity -> (1, 0, 0, 0)


def test_counter_summaries_are_keyed_by_the_kernel_sources_and_stale_ones_are_labelled(tmp_path, monkeypatch):
    from r2s_hip._lib import kernel_source_sha16

    sha = kernel_source_sha16()
    assert len(sha) == 16 and sha == kernel_source_sha16()
    sys.path.insert(0, ROOT)
    import bench

    committed = json.load(open(os.path.join(ROOT, bench.PMC_FILE)))
    assert "source_sha16" in committed and "k_substep" in committed["sloth_32env"] and "k_composite" in committed["sloth_32env"]
    assert committed["sloth_32env"]["k_composite"]["valu_busy_frac"] <= 1.0 and committed["sloth_32env"]["k_substep"]["valu_busy_frac"] <= 1.0
    ent, src = bench.pmc_summary("k_substep", "sloth_32env")
    assert ent is not None and ent["stale"] == (committed["source_sha16"] != sha) and ("STALE" in src) == ent["stale"]
    # a summary collected on other sources must be labelled, whatever it says
    fake = dict(committed, source_sha16="0" * 16)
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps(fake))
    monkeypatch.setattr(bench, "PMC_FILE", str(p))
    monkeypatch.setattr(bench, "ROOT", "/")
    ent, src = bench.pmc_summary("k_substep", "sloth_32env")
    assert ent["stale"] is True and "STALE" in src
    assert bench.pmc_summary("no_such_kernel", "sloth_32env") == (None, None)


def test_folded_rope_scenes_rest_without_strain_and_without_resting_pairs_between_the_legs():
    """`rope_fold` / `rope_tip_fold` (r2s_hip/synth.py: the scenes of the resident stepper's self-collision flavour): the hairpin is the
    REST shape — springs are built on it, none joins the two legs — and the legs are about as far apart as the reference's resting-pair
    marking reaches (spring_mass_warp.py:272-291), so nearly every pair of particles of the two legs may become a candidate once gravity
    has laid the upper leg on the lower one (tests/test_physics_oracle_kat.py counts them: more than 500 of 2 080 particles)."""
    from scipy.spatial import cKDTree
    from r2s_hip import synth
    from r2s_hip.rollout import CONFIGS

    assert CONFIGS["rope_fold_1env"][0] == "rope_fold" and CONFIGS["rope_tip_fold_1env"][0] == "rope_tip_fold"
    for shape, upper_min, upper_max in (("rope_fold", 0.3, 0.6), ("rope_tip_fold", 0.05, 0.11)):
        ob = synth.phystwin_object(shape, 2000, 0)
        p = ob["points"].astype(np.float64)
        rest = np.linalg.norm(p[ob["springs"][:, 0]] - p[ob["springs"][:, 1]], axis=1)
        assert np.abs(rest - ob["rest"]).max() < 1e-6                          # no strain at rest
        z_mid = 0.5 * (p[:, 2].min() + p[:, 2].max())
        x_bend = p[:, 0].max() - 0.03                                          # the half circle: the last 2.5 cm in x
        lower = (p[:, 2] < z_mid) & (p[:, 0] < x_bend)
        upper = (p[:, 2] > z_mid) & (p[:, 0] < x_bend)
        assert lower.sum() > 500 and upper.sum() > 50
        span = p[upper, 0].max() - p[upper, 0].min()
        assert upper_min < span < upper_max, span
        d, _ = cKDTree(p[lower]).query(p[upper])
        assert d.min() > 0.022, d.min()                                        # 26 mm between the surfaces less the lattice's jitter: about the
                                                                               # resting-pair reach (5 x collision_dist; cell-granular, :287-291)
        s = ob["springs"]
        assert not ((lower[s[:, 0]] & upper[s[:, 1]]) | (upper[s[:, 0]] & lower[s[:, 1]])).any()
