"""bench.py's `cpu_baseline` leg: the environments x particle-chunks driver of the oracle
(r2s_oracle_phys_step_batch_par_f32) and the optimised build libr2s_cpu_baseline.so it is timed from.  The driver calls the SAME
per-particle functions as the sequential stepper over ranges, with eval_springs as a gather in the scatter's summation order, so
under the checker's flags its positions are bit-equal; the optimised build (FMA contraction, -O3) stays within the 1e-5 gate."""
import numpy as np

import oracle
from util_physics import gripper_motion, make_object, oracle_env, two_blobs


def _scene(self_collision):
    from r2s_hip import synth

    if self_collision:
        ob = two_blobs(seed=3, gap=0.06, speed=1.5, n=120)       # built apart (resting pairs exclude what is close at construction)
    else:
        ob = make_object("sloth", 400, seed=5, lift=0.0005)
    c, top = ob["points"].mean(0), ob["points"][:, 2].max()
    fingers = [synth.finger_mesh((c[0], c[1] - 0.02, top - 0.01)), synth.finger_mesh((c[0], c[1] + 0.02, top - 0.01))]
    box = synth.box_mesh((c[0] + 0.2, c[1], 0.05), (0.1, 0.1, 0.1))
    return ob, fingers, box


def _envs(ob, fingers, box, n, n_sub, self_collision):
    envs = []
    interp, centers, dv, om = gripper_motion(fingers, n_sub, 5e-5, vel=(0.0, 0.0, -0.5), closing=0.4)
    for e in range(n):
        o = dict(ob); o["points"] = ob["points"] + np.float32(0.01 * e) * np.array([1, 0, 0], np.float32)
        fe = [(v + np.float32(0.01 * e) * np.array([1, 0, 0], np.float32), f) for v, f in fingers]
        env = oracle_env(o, num_substeps=n_sub, dynamic_meshes=fe, static_meshes=[box], self_collision=self_collision)
        env.set_mesh_interactive(interp + np.float32(0.01 * e) * np.array([1, 0, 0], np.float32), centers, dv, om)
        if self_collision:
            nA = int(np.flatnonzero(o["v0"][:, 0] != 0)[0])
            env.x[nA:, 0] -= np.float32(0.0565)                    # ... then moved to 3.5 mm from the first blob: live candidates
            env.update_collision_graph()
        envs.append(env)
    return envs


def test_env_x_chunk_driver_equals_the_sequential_stepper_bit_for_bit():
    n_sub = 60
    for sc in (False, True):
        ob, fingers, box = _scene(sc)
        a, b = _envs(ob, fingers, box, 3, n_sub, sc), _envs(ob, fingers, box, 3, n_sub, sc)
        oracle.phys_step_batch(a, n_sub)
        ran = oracle.phys_step_batch_par(b, n_sub, threads_per_env=4)
        assert ran >= 3
        touched = False
        for ea, eb in zip(a, b):
            assert np.array_equal(ea.x, eb.x) and np.array_equal(ea.v, eb.v), float(np.abs(ea.x - eb.x).max())
            assert np.allclose(ea.collision_forces, eb.collision_forces, rtol=1e-5, atol=1e-2)   # atomic adds: order only
            touched = touched or float(np.abs(ea.collision_forces).max()) > 0
        assert touched, "the fingers must reach the object in this scenario"
        if sc:
            assert max(int((e.coll_num > 0).sum()) for e in a) > 0, "live self-collision candidates"


def test_optimised_baseline_build_stays_inside_the_position_gate():
    n_sub = 60
    ob, fingers, box = _scene(True)
    a, b = _envs(ob, fingers, box, 2, n_sub, True), _envs(ob, fingers, box, 2, n_sub, True)
    oracle.phys_step_batch(a, n_sub)
    with oracle.baseline_build():
        assert oracle.lib() is not None
        oracle.phys_step_batch_par(b, n_sub, threads_per_env=2)
    assert oracle.lib()._name.endswith("libr2s_oracle.so"), "the checker build is back after the block"
    for ea, eb in zip(a, b):
        assert float(np.abs(ea.x - eb.x).max()) < 1e-5


def test_driver_without_meshes_or_self_collision_has_its_own_barrier():
    """Advisor r4: with no mesh and no self-collision nothing separated a thread's gather (reads its neighbours' x, v) from another
    thread's integrate_ground (overwrites them in place).  Many threads over a small object make a missing barrier show."""
    n_sub = 40
    ob = make_object("rope", 300, seed=2, lift=0.001)
    a = [oracle_env(ob, num_substeps=n_sub, self_collision=False) for _ in range(2)]
    b = [oracle_env(ob, num_substeps=n_sub, self_collision=False) for _ in range(2)]
    oracle.phys_step_batch(a, n_sub)
    for _ in range(3):
        c = [oracle_env(ob, num_substeps=n_sub, self_collision=False) for _ in range(2)]
        oracle.phys_step_batch_par(c, n_sub, threads_per_env=8)
        for ea, ec in zip(a, c):
            assert np.array_equal(ea.x, ec.x) and np.array_equal(ea.v, ec.v)
    del b
