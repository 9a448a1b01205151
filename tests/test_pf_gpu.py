"""Round 5: the contact flavours of a large batch with the finishing code of substep k at the HEAD of substep k + 1's launch
(k_substep_pf; csrc/physics_substep.h "finishing at the HEAD of the next launch", csrc/physics_finish.h) against the two-launch form (k_substep + k_contact_finish
per substep, r2s_phys_set_pf(h, 0)).  Same arithmetic on the same inputs in the same order, another transport (a tagged write-through
result line per particle instead of the state array + a launch boundary): the states must agree BIT FOR BIT, every env step, with
deferred mesh queries, tagged entries and live self-collision candidates in play — and two runs of the same rollout, enqueued without
a host synchronisation in between, must too (the flavour of step t follows from the counters of step t - 2, waited for: VERDICT r4
item 4).  The oracle-level parity of this flavour is what the rest of the GPU suite checks: it is the default."""
import numpy as np
import pytest

from util_parity import record

pytestmark = pytest.mark.gpu


def _rollout(cfg, n_env, pf, steps, close_at=2, sync_every_step=False, busy=False, **kw):
    import torch
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout(cfg, n_env=n_env, close_at=close_at, **kw)
    assert ro.phys.layout_stats()["lds_bytes"] == 1024 * 24, "the large-batch layout"
    ro.phys.set_pf(pf)
    xs, vs, fl = [], [], []
    side = torch.cuda.Stream() if busy else None
    big = torch.empty(1 << 27, dtype=torch.float32, device="cuda") if busy else None      # 512 MiB
    for _ in range(steps):
        if busy:                     # a second stream keeps every CU streaming through HBM next to the hand-offs (uneven load)
            with torch.cuda.stream(side):
                for _ in range(6):
                    big.mul_(1.0001)
        ro.physics_step()
        ro.t += 1
        if sync_every_step:
            torch.cuda.synchronize()
        xs.append(ro.phys.x.clone()); vs.append(ro.phys.v.clone()); fl.append(ro.phys.last_flavour())
    torch.cuda.synchronize()
    st = ro.contact_stats()
    ro.phys.step(0, 0)            # a sticky fault (a poll that hit its limit) would raise here
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in xs], [v.cpu().numpy() for v in vs], fl, st, ro.phys.deferred_counts(), ro.phys.tagged_count()


def test_grasp_of_the_toy_two_chains_finishers_at_head_equal_two_launches_bit_for_bit():
    """9 environments of the headline scene (two concurrent chains), the gripper closes on the toy's arms at env step 2: finger contact
    (deferred queries with the triangles in registers), the arms pressed together (candidates; tagged entries where both meet)."""
    steps = 6
    xa, va, fa, sa, da, ta = _rollout("sloth_32env", 9, True, steps)
    xb, vb, fb, sb, db, tb = _rollout("sloth_32env", 9, False, steps)
    assert [f["deferred_mesh_queries"] for f in fa] == [f["deferred_mesh_queries"] for f in fb]
    assert [f["self_collision_kernel"] for f in fa] == [f["self_collision_kernel"] for f in fb]
    assert any(f["finishers_at_head_of_next_launch"] for f in fa) and not any(f["finishers_at_head_of_next_launch"] for f in fb), [f["kernel"] for f in fa]
    assert fa[-1]["finishers_at_head_of_next_launch"] and fa[-1]["self_collision_kernel"] and fa[-1]["chains"] == 2, fa[-1]
    assert sa["mesh_contacts"] > 0 and sa["self_collision_candidates"] > 0 and da[:-1].max() > 0, (sa, da.max())
    assert {k: v for k, v in sa.items() if k != "flavour"} == {k: v for k, v in sb.items() if k != "flavour"} and np.array_equal(da, db) and ta == tb
    for k in range(steps):
        assert np.array_equal(xa[k], xb[k]) and np.array_equal(va[k], vb[k]), (k, float(np.abs(xa[k] - xb[k]).max()))
    assert np.isfinite(xa[-1]).all()
    record("grasp of the toy, finishers at the head of the next launch vs two launches", env_steps=steps, envs=9, mesh_contacts=sa["mesh_contacts"],
           candidates=sa["self_collision_candidates"], tagged=ta, x_max_abs=0.0, tol=0)


def test_pusher_rod_against_the_block_large_mesh_records_per_environment():
    """9 environments of the push-T scene: the 25k-face rod reaches the block at env step 2 (per-environment records, box hierarchy,
    four wavefronts per listed particle at the head of the launch)."""
    steps = 5
    xa, va, fa, sa, da, _ = _rollout("T_pusher_32env", 9, True, steps)
    xb, vb, fb, sb, db, _ = _rollout("T_pusher_32env", 9, False, steps)
    assert all(f["finishers_at_head_of_next_launch"] and f["mesh_template"] == 2 for f in fa) and not any(f["finishers_at_head_of_next_launch"] for f in fb)
    assert sa["mesh_contacts"] > 0 and da[:-1].max() > 0 and np.array_equal(da, db), (sa, da.max())
    for k in range(steps):
        assert np.array_equal(xa[k], xb[k]) and np.array_equal(va[k], vb[k]), (k, float(np.abs(xa[k] - xb[k]).max()))
    record("pusher rod against the block, finishers at the head of the next launch vs two launches", env_steps=steps, envs=9,
           mesh_contacts=sa["mesh_contacts"], x_max_abs=0.0, tol=0)


@pytest.mark.parametrize("cfg,n_env", [("sloth_32env", 9), ("rope_1env", 1)])
def test_two_identical_rollouts_through_a_grasp_without_host_synchronisation_are_bit_identical(cfg, n_env):
    """the flavour of an env step must not depend on when a device-to-host copy happens to land.  The same rollout
    twice — once enqueued as fast as the host can (the GPU runs several steps behind), once with a synchronisation after every step —
    must run the same flavours and end in the same bits."""
    steps = 7
    xa, va, fa, *_ = _rollout(cfg, n_env, True, steps) if n_env > 1 else _rollout_small(cfg, steps, False)
    xb, vb, fb, *_ = _rollout(cfg, n_env, True, steps, sync_every_step=True) if n_env > 1 else _rollout_small(cfg, steps, True)
    assert [f["kernel"] for f in fa] == [f["kernel"] for f in fb], ([f["kernel"] for f in fa], [f["kernel"] for f in fb])
    for k in range(steps):
        assert np.array_equal(xa[k], xb[k]) and np.array_equal(va[k], vb[k]), (k, float(np.abs(xa[k] - xb[k]).max()))


def _rollout_small(cfg, steps, sync_every_step):
    import torch
    from r2s_hip.rollout import BatchedRollout

    ro = BatchedRollout(cfg, close_at=2)
    xs, vs, fl = [], [], []
    for _ in range(steps):
        ro.physics_step()
        ro.t += 1
        if sync_every_step:
            torch.cuda.synchronize()
        xs.append(ro.phys.x.clone()); vs.append(ro.phys.v.clone()); fl.append(ro.phys.last_flavour())
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in xs], [v.cpu().numpy() for v in vs], fl


def test_hand_offs_under_uneven_load_a_second_stream_streaming_through_hbm():
    """The result lines are a cross-workgroup (cross-XCD) hand-off inside a launch: tested where such hand-offs fail — under uneven load,
    with the pollers' lines warm (cdna_hip_programming.md, Guideline 16).  The same 9-environment grasp on a quiet chip and next to a stream
    that keeps rewriting 512 MiB: every word of every state equal, no poll ran into its limit."""
    steps = 6
    xa, va, fa, sa, *_ = _rollout("sloth_32env", 9, True, steps)
    xb, vb, fb, sb, *_ = _rollout("sloth_32env", 9, True, steps, busy=True)
    assert [f["kernel"] for f in fa] == [f["kernel"] for f in fb] and fa[-1]["finishers_at_head_of_next_launch"]
    assert sa["mesh_contacts"] > 0 and sa["self_collision_candidates"] > 0
    for k in range(steps):
        assert np.array_equal(xa[k], xb[k]) and np.array_equal(va[k], vb[k]), (k, float(np.abs(xa[k] - xb[k]).max()))


def test_chains_captured_as_head_and_tail_graphs_equal_one_graph_per_chain_bit_for_bit():
    """Round 6: every chain is captured as a head graph (R2S_GRAPH_HEAD substeps, default 64) and a tail graph, all heads launched before any
    tail (csrc/physics.hip capture_graph / launch_graphs) — the same launches in the same order on the same streams.  The grasp of the toy
    (two chains, finishers at the head of the next launch, candidates) with one graph per chain (0), the default, and a head of 7 substeps
    (odd: the tail starts on the other parity of the state buffer): every env step the same bits."""
    import os

    steps = 5
    runs = {}
    old = os.environ.get("R2S_GRAPH_HEAD")
    try:
        for head in ("0", "64", "7"):
            os.environ["R2S_GRAPH_HEAD"] = head          # read once per handle, at create
            runs[head] = _rollout("sloth_32env", 9, True, steps)
    finally:
        if old is None:
            os.environ.pop("R2S_GRAPH_HEAD", None)
        else:
            os.environ["R2S_GRAPH_HEAD"] = old
    xa, va, fa, sa, da, ta = runs["0"]
    assert fa[-1]["finishers_at_head_of_next_launch"] and fa[-1]["chains"] == 2 and sa["mesh_contacts"] > 0 and sa["self_collision_candidates"] > 0
    for head in ("64", "7"):
        xb, vb, fb, sb, db, tb = runs[head]
        assert [f["kernel"] for f in fa] == [f["kernel"] for f in fb]
        assert np.array_equal(da, db) and ta == tb
        for k in range(steps):
            assert np.array_equal(xa[k], xb[k]) and np.array_equal(va[k], vb[k]), (head, k, float(np.abs(xa[k] - xb[k]).max()))
    record("chains as head + tail graphs vs one graph per chain", env_steps=steps, envs=9, heads=[0, 64, 7], x_max_abs=0.0, tol=0)
