"""The reference's control loop, closed on the CPU (oracle/closed_loop.py: PhysOracle's last-substep forces -> EefOracle's grasp state
machine -> finger vertices -> PhysOracle; phystwin.py:362-521), on a scene small enough for the CPU suite: a 600-particle rope under the
two-finger gripper.  What it pins:

  * with the commanded opening RAMPED down (a policy's "close"), both pads' filtered forces (faces 18, 19, 1 of each finger,
    phystwin.py:386-392) exceed grasp_force_threshold = 3e4 while the command is still below the current opening, `grasped` latches,
    the opening FREEZES above the command, the rope is carried up by the lift, and the re-opening command releases it (< 100 on both
    pads) — every branch of phystwin.py:394-408 runs from the stepper's own forces, none scripted;
  * with the command JUMPING to its closed value within one env step (rounds 1-5 of this repo) the state machine can never latch:
    from the next step on the command equals the current opening (`grasped_envs` was 0 in every bench line).

The same loop against the HIP rollout: tests/test_grasp_closed_loop_gpu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real2sim-eval_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _rollout(close_rate, steps, close_at=2, open_at=10**9):
    from oracle.closed_loop import OracleRollout
    from r2s_hip.rollout import scene_setup

    scn = scene_setup("tiny", seed=0, schedule="grasp", close_at=close_at, open_at=open_at, close_rate=close_rate)
    ro = OracleRollout(scn, threads=min(8, os.cpu_count() or 1))
    z0 = float(ro.x[:, 2].max())
    for _ in range(steps):
        ro.step()
    return ro, z0


def test_pad_faces_are_the_gripping_flats_of_a_closed_finger_mesh():
    """The stand-in finger has the reference meshes' topology (left / right_finger_large_2.stl: 44 faces on 24 welded vertices, every edge
    shared by exactly two faces, once in each direction) and its gripping flat sits where the grasp test reads: faces 18, 19 (+ 1)."""
    from collections import Counter

    from r2s_hip import synth

    tab, init, fl, fr = synth.gripper_eef_table()
    assert tab.shape == (101, 48, 3) and fl.shape == fr.shape == (44, 3)
    for pad_normal in ((0.0, 1.0, 0.0), (0.0, -1.0, 0.0)):
        v, f = synth.finger_mesh((0.0, 0.0, 0.0), pad_normal=pad_normal)
        assert v.shape == (24, 3) and f.shape == (44, 3)
        e = Counter((int(t[k]), int(t[(k + 1) % 3])) for t in f for k in range(3))
        assert all(c == 1 for c in e.values()) and all((b, a) in e for a, b in e), "closed, consistently oriented manifold"
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        n = np.cross(b - a, c - a)
        area = 0.5 * np.linalg.norm(n, axis=1)
        n /= 2 * area[:, None]
        assert np.allclose(n[18], pad_normal, atol=1e-5) and np.allclose(n[19], pad_normal, atol=1e-5)
        assert n[1] @ np.asarray(pad_normal) > 0.7 and abs(n[1][2]) < 1e-6            # a chamfer next to the pad
        assert np.isclose(area[18] + area[19], (0.02 - 0.002) * (0.05 - 0.002), rtol=1e-4)   # the whole flat
        assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() > 0                  # outward


def test_grasp_latches_holds_lifts_and_releases_from_the_steppers_own_forces():
    ro, z0 = _rollout(0.1, 26, close_at=2, open_at=20)
    log = ro.log
    g = [l["grasped"] for l in log]
    assert any(g), [(l["t"], l["command"], l["force_in"]) for l in log]
    t_g = g.index(True)
    # latched while the command was still falling, from forces above the threshold on BOTH pads
    assert 2 + 4 <= t_g <= 2 + 11 and min(log[t_g]["force_in"]) > 3e4 and log[t_g]["command"] < log[t_g]["openness"], log[t_g]
    # ... and before that the opening followed the command down (the closing branch)
    assert all(abs(l["openness"] - l["command"]) < 1e-6 for l in log[:t_g])   # (the command passes through float32 like gripper_openness.item())
    # held: the opening stays frozen (or creeps by at most 0.05 per step while a pad's force sags) although the command goes on to 0
    held = log[t_g:20]
    assert all(l["grasped"] for l in held) and held[-1]["command"] == 0.0 and held[-1]["openness"] >= 0.05
    assert all(-1e-9 <= a["openness"] - b["openness"] <= 0.05 + 1e-9 for a, b in zip(held, held[1:]))
    # lifted in the grasp: the rope's top follows the end effector up
    lift_steps = 20 - (2 + 10)
    assert ro.scn["close_at"] + 10 <= 20 and lift_steps > 0
    # released: the command jumps to 1, the pads leave the rope, both forces fall below 100 and `grasped` drops
    assert log[20]["command"] == 1.0 and log[20]["openness"] == 1.0
    assert not log[-1]["grasped"] and max(log[-1]["force_in"]) < 100.0, log[-1]


def test_rope_rises_with_the_gripper_while_it_is_held():
    ro, _ = _rollout(0.1, 12, close_at=2)            # the closing ramp ends with step 11: the lift starts at step 12
    assert ro.log[-1]["grasped"]
    c = ro.scn["ob"]["points"].mean(0)
    between = (np.abs(ro.x[:, 0] - c[0]) < 0.01)      # the 2 cm of rope between the pads
    z_a = float(ro.x[between, 2].mean())
    for _ in range(7):
        ro.step()
    assert ro.log[-1]["grasped"]
    # 7 steps of lift at 5 cm/s = 1.2 cm of end-effector travel: the gripped stretch of the rope comes up with it (friction 1.0 on the pads)
    z_b = float(ro.x[between, 2].mean())
    assert z_b > z_a + 0.008, (z_a, z_b)


def test_a_jumping_command_can_never_latch_the_grasp():
    ro, _ = _rollout(None, 12, close_at=2)
    assert not any(l["grasped"] for l in ro.log)
    # ... although the pads do load up beyond the threshold: it is the schedule, not the forces, that kept rounds 3-5 at grasped_envs = 0
    assert max(min(l["force_in"]) for l in ro.log) > 3e4
