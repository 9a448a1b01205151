"""bench.py keeps the driver's contract: one JSON line with the agreed keys, roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "2", "--warmup", "1", "--substeps", "30",
                          "--cpu-budget", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k, t in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float, higher_is_better=bool, scaling=str,
                     dtype=str, data=str, config=dict, roofline=dict, cpu_baseline=dict).items():
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"]
    assert d["unit"] == "env-steps/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "env-steps/s" and isinstance(c["sample"], str) and c["value"] > 0
    assert abs(d["value"] - d["config"]["envs_per_gpu"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert set(d["phases"]) >= {"free", "contact", "note"} and d["phases"]["free"]["steps"] >= 1
    for k in ("traffic_source", "hbm_actual_frac", "valu_busy_frac", "algorithmic_bytes_per_launch", "avg_launch_us", "frac_shared_topology",
              "traffic_over_shared", "bound_in_practice"):
        assert k in r, k
    assert r["frac_shared_topology"] <= r["frac"] + 1e-12
    assert c["threads"]["physics"] == c["cores"] and "libr2s_cpu_baseline" in c["sample"]
    g = d["parity_gate"]
    assert g["passed"] and "flavour" in g and "threshold_flip_pixels" in g and g["hard_rgb_mismatch_pixels"] == 0, g
    cl = d["closed_loop_get_obs"]
    assert cl["env_steps_per_s"] > 0 and 0 < cl["ratio"] <= 1.05 and "re_rendered_batches" in cl, cl


def test_headline_schedule_contains_free_motion_finger_contact_and_live_self_collision():
    """the timed window of the headline workload must not be a contact-free best case.  A 2-env cut of
    configs[2] (same object, same 667 substeps, same action trace): the window starts in free motion; from the second step of the
    closing ramp (round 6: the commanded opening falls by 0.1 per step from six steps before the window's middle, so that the grasp
    latches from the stepper's own forces inside the window) particles are inside the fingers' collision margin, then live
    self-collision candidates (the toy's arms pressed together) and the graph flavour with the self-collision variant of the fused
    kernel plus the finishing code, and the grasp state machine latches: the window's last steps are the held grasp."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "2", "--steps", "16", "--warmup", "2", "--no-cpu-baseline", "--episodes", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][-1])
    free, contact = d["phases"]["free"], d["phases"]["contact"]
    assert free["steps"] == 3 and contact["steps"] == 13            # the ramp starts at timed step 16 / 2 - 6 = 2, the pads arrive one step later
    assert contact["grasped_envs"] == 2, contact                    # ... and the grasp latches from the stepper's own forces
    assert free["mesh_contacts"] == 0 and free["self_collision_candidates"] == 0
    assert contact["mesh_contacts"] > 0 and contact["self_collision_candidates"] > 0, contact
    # SELF=true + finishing code (round 5: at the head of the next substep's launch, k_substep_pf; before: k_contact_finish as a launch of its own)
    assert any(",true," in f and ("k_contact_finish" in f or "k_substep_pf" in f) for f in contact["kernel_flavours"]), contact["kernel_flavours"]
    assert all(",true," not in f for f in free["kernel_flavours"]), free["kernel_flavours"]
    assert "sloth_32env" in d["config"]["workload"] and "grasp" in d["config"]["workload"]
