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
