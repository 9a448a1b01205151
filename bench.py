#!/usr/bin/env python
"""bench.py — env-steps/s (physics + render) of the batched rollout on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-launches itself as N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one batched env step of the hot path over `envs` environments per GPU: collision-candidate rebuild, gripper
kinematics + grasp logic, 667 fused physics substeps, skinning, and 2 rasterised 640x480 frames per environment
(SURVEY.md §8d Metric 1), on synthetic inputs of BASELINE.json configs[2] (sloth PhysTwin ~15k particles / ~80k Gaussians,
32 envs per GPU).  Inputs are resident in HBM before the timed region.  Envs shard across GPUs with no data-path
collective (weak scaling); the only collective is the final all-gather of per-rank result records.

The timed window is what a POLICY IN THE LOOP sees (round 5; eval_policy.py:124-213): every step ends in get_obs() — the host
waits for the frames, validates the sync-free raster batch, re-renders a lossy one — before the next action is applied, and the
environments are DE-PHASED: environment e runs the action trace (e K/2) // E steps late, so the grippers close one after the other
(episodes of eval_policy_parallel.py do not share a phase).  Round 6: the gripper CLOSES LIKE A POLICY CLOSES IT — the commanded opening
ramps down by --close-rate per env step (default 0.1) towards 0 instead of jumping — so the reference's grasp state machine
(phystwin.py:383-412) runs as it does in an episode: the pads load up step by step, both filtered pad forces exceed 3e4, `grasped`
latches, the opening freezes (or creeps by 0.05 per step while a force sags), and the object is LIFTED in the grasp.  Until round 5 the
command jumped to its closed value within one env step, after which it can never again be below the current opening — the condition
under which the reference establishes a grasp — and `grasped_envs` was 0 in every bench line of rounds 3-5.  The window starts with
the first grippers already closing (the trace's closing step is `warmup - 4`) and the delays cover K/2 steps: over the K timed steps
some environments are still coming down or closing (free motion, first finger contact) and from about two thirds of the window on all
of them hold the toy's arms and lift (`window.grasped_envs_per_step` ends at the batch size) — finger contact on every pad, the arms
pressed together (live self-collision candidates), the state machine in its hold / creep branches.  `value` is the mean over that
window; `window.step_latency_ms` says what a single step of it costs at most.  Reported next to it, never as `value`: the same K steps
with all environments in phase (`synchronised_window`; `phases` splits it into free motion and contact), enqueue-only, with the
rasterisation of step t next to the substeps of step t+1, and SUSTAINED full episodes (`episodes`: reset -> 30 settling steps ->
450 policy steps on every slot, observation sink on; eval_policy.py:65-267).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the fused physics substep) and `cpu_baseline`
(the oracle — a CPU restatement of the reference algorithm, kind "port" — on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "real2sim-eval_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_FILE = next((f for f in (os.path.join("profiles", "r6_pmc_summary.json"), os.path.join("profiles", "r5_pmc_summary.json"), os.path.join("profiles", "r4_pmc_summary.json")) if os.path.exists(os.path.join(ROOT, f))),
                os.path.join("profiles", "r6_pmc_summary.json"))   # the newest committed counter summary (labelled STALE when collected on other kernel sources)


def pmc_summary(kernel, config):
    """Counter-derived figures of `kernel` from the committed rocprofv3 --pmc passes over THIS workload (profiles/, produced
    by tools/profiling/pmc_r4.sh; FETCH_SIZE x2 + WRITE_SIZE per the microarchitecture guide).  They are NOT measured in this
    run — the JSON says so next to every number taken from here — and the summary names the kernel sources it was collected
    on (`source_sha16`, r2s_hip._lib.kernel_source_sha16): a summary of OTHER sources is labelled stale.
    Returns (entry or None, provenance string)."""
    try:
        from r2s_hip._lib import kernel_source_sha16
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
        ent = d[config][kernel]
        same = d.get("source_sha16") == kernel_source_sha16()
        src = (f"{PMC_FILE} (rocprofv3 --pmc passes of this workload, committed; NOT measured in this run; collected on "
               + ("these kernel sources" if same else "OTHER kernel sources: STALE") + f", git {d.get('git_head', '?')[:12]})")
        return dict(ent, stale=not same), src
    except Exception:
        return None, None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ro, budget_s=20.0):
    """Time the CPU restatement on a bounded sample of the same workload, built the way a CPU user would build it
    (oracle/libr2s_cpu_baseline.so: the oracle's sources with -O3, AVX2 + FMA code generation, FMA contraction — BASELINE.md §2;
    the strict-IEEE checker build is NOT what is timed) and parallelised over environments x particle chunks
    (r2s_oracle_phys_step_batch_par_f32) so that every core the process may use is busy: `n` environments side by side, each with
    a team of cores // n threads, self-collision ON (candidate rebuild once per env step + object_collision every substep, like
    the workload), meshes + ground; then the frames of one environment with all threads over tiles.  Extrapolated to env-steps/s."""
    import concurrent.futures as cf

    import oracle

    cores = len(os.sched_getaffinity(0))
    n = max(1, min(ro.n_env, cores))
    tpe = max(1, min(cores // n, 32))
    with oracle.baseline_build():
        oracle.set_threads(cores)
        dyn = ro.fingers if ro.with_gripper else None
        x0 = ro.ob["points"]
        sta = None
        if ro.phys.n_faces and (ro.phys.mesh_map < 0).any():
            from r2s_hip import synth
            c = x0.mean(0)
            sta = [synth.box_mesh((c[0] + 0.25, c[1] + 0.2, 0.135), (0.2, 0.13, 0.27))]
        sc = bool(ro.phys.self_collision)

        def make(e):
            return oracle.PhysOracle(x0 + ro.env_shift[e], ro.ob["springs"], ro.ob["rest"], ro.ob["log_Y"], num_substeps=ro.num_substeps,
                                     self_collision=sc, dynamic_meshes=dyn, static_meshes=sta, use_pusher=ro.use_pusher)

        with cf.ThreadPoolExecutor(max_workers=n) as pool:        # ctypes calls release the GIL: construction (resting pairs) in parallel
            envs = list(pool.map(make, range(n)))
            t_rebuild = 0.0
            if sc:                                                  # update_collision_graph: once per env step, environments side by side
                t0 = time.perf_counter()
                list(pool.map(lambda o: o.update_collision_graph(), envs))
                t_rebuild = time.perf_counter() - t0
        # physics: calibrate on 2 substeps, then spend ~60% of the budget
        t0 = time.perf_counter(); ran = oracle.phys_step_batch_par(envs, 2, tpe); t1 = time.perf_counter()
        per = max((t1 - t0) / 2, 1e-5)
        nsub = int(max(4, min(ro.num_substeps, 0.6 * budget_s / per)))
        t0 = time.perf_counter(); ran = oracle.phys_step_batch_par(envs, nsub, tpe); t_phys = (time.perf_counter() - t0) / nsub
        # raster: frames side by side (environments x cameras), a small team over the tiles of each — one frame's pipeline is partly
        # sequential (duplicate + sort), so one frame at a time would leave most of the cores idle
        tpf = 4
        n_par = max(1, min(n * len(ro.cams), cores // tpf))
        jobs = [(e, v) for e in range(n) for v in range(len(ro.cams))][:n_par]
        scenes = {e: (ro.means[e].cpu().numpy(), {k: t.cpu().numpy() for k, t in ro.g_env(e).items()}) for e in {e for e, _ in jobs}}
        cams = {(e, v): ro.camera_numpy(e, v) for e, v in jobs}

        def frame(job):
            oracle.set_threads(tpf)                          # the OpenMP team size is a per-thread setting: this worker's frames only
            means, g = scenes[job[0]]
            cam = cams[job]
            oracle.raster_forward(means, g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"], cam["tanfovy"],
                                  ro.H, ro.W, cam["bg"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"], z_threshold=cam["z_threshold"])

        with cf.ThreadPoolExecutor(max_workers=n_par) as pool:
            t0 = time.perf_counter()
            list(pool.map(frame, jobs))
            t_frames = time.perf_counter() - t0          # n_par frames, n_par x tpf threads
        frame_rate = n_par / t_frames
    # n envs in parallel for physics; their frames at the measured frame rate
    t_env_step_batch = t_rebuild + t_phys * ro.num_substeps + n * len(ro.cams) / frame_rate
    return {
        "value": n / t_env_step_batch, "unit": "env-steps/s", "cores": int(max(ran, 1)), "kind": "port",
        "sample": f"physics: {n} envs x {tpe} threads = {ran} threads busy (envs x particle chunks), {nsub} of {ro.num_substeps} substeps, springs + "
                  f"{'self-collision (rebuild once per env step + object_collision per substep) + ' if sc else ''}meshes + ground; raster: {n_par} frames ({ro.W}x{ro.H}, environments x cameras) side by side, "
                  f"{tpf} threads over the tiles of each = {n_par * tpf} threads; extrapolated to full env steps; build: oracle/libr2s_cpu_baseline.so (gcc -O3 "
                  f"-march=x86-64-v3, FMA contraction) — the CPU restatement of the reference algorithm, not upstream code (the reference has no CPU path)",
        "threads": {"physics": int(ran), "raster": int(n_par * tpf), "available": int(cores)},
        "phys_ms_per_substep_batch": t_phys * 1e3, "candidate_rebuild_ms_per_env_step_batch": t_rebuild * 1e3,
        "raster_ms_per_frame_amortised": 1e3 / frame_rate, "cpu_threads_available": cores, "cpu_model": _cpu_model(),
    }


def _pct(v, q):
    v = sorted(v)
    return float(v[min(len(v) - 1, int(round(q * (len(v) - 1))))]) if v else None


def latency_stats(ms):
    return {"p50": _pct(ms, 0.5), "p99": _pct(ms, 0.99), "max": max(ms) if ms else None, "mean": sum(ms) / len(ms) if ms else None, "steps": len(ms)}


EPISODE_STEPS = {"sloth": 450, "rope": 900, "T": 1800}   # env.sim.duration x 30 fps: scripts/eval_policy/sloth_act.sh (15 s), cfg/env/xarm_gripper.yaml:3 (30 s), xarm_pusher.yaml:3 (60 s)


def sustained_episodes(config, dev, seed, n_env, substeps, res, per_slot, steps, close_rate, sink_dir):
    """FULL episodes on every environment slot through the episode scheduler (r2s_hip/evaluate.run_episodes: eval_policy.py:65-267 for a
    batch): per-slot reset into the episode's randomised start pose -> 30 settling steps -> `steps` steps of the scene's action trace
    (approach, closing ramp, grasp, lift, hold / the rod pushing the block), get_obs() before every step, the observation sink writing
    every frame and state (JPEG + pickle per environment and step, worker processes).  Never `value`: the sustained figure NEXT to the
    20-step window — tails a short mean hides (flavour switches, resets, re-rendered batches, sink stalls) show up here."""
    import shutil
    import tempfile

    import torch
    from r2s_hip.evaluate import run_episodes, summarize
    from r2s_hip.rollout import BatchedRollout
    from r2s_hip.sink import ObservationSink

    settle = 30
    kw = dict(device=dev, seed=seed, n_env=n_env, num_substeps=substeps, res=res, close_at=settle + 15, randomize="multicam" not in config)
    if close_rate and "pusher" not in config:
        kw.update(close_rate=close_rate, lift_steps=60)
    ro = BatchedRollout(config, **kw)
    steps = int(steps or EPISODE_STEPS.get(ro.ob_shape, 450))
    E = ro.n_env
    tmp = sink_dir or tempfile.mkdtemp(prefix="r2s_sink_")
    sink = ObservationSink(tmp, E, ro.views, ro.H, ro.W, device=dev, slots=6, state_bytes=2 * E * ro.N * 12 + 4096)
    lat, grasped_trace, flav = [], [], set()
    last = [time.perf_counter()]
    cnt = [0]
    gdev = torch.zeros(steps + settle + 2, 3, dtype=torch.int32, device=dev)

    def on_step(r, slot_episode, step_t):
        sink.submit(cnt[0] % (steps + settle), r.out_color, state=dict(x=r.phys.x, v=r.phys.v))
        if cnt[0] < gdev.shape[0]:
            r.phys.log_contacts(gdev[cnt[0]])          # {candidates, mesh hits, grasped environments} of this step, kept on the device
        if cnt[0] % 16 == 0:
            flav.add(r.phys.last_flavour()["kernel"])
        cnt[0] += 1
        now = time.perf_counter()
        lat.append((now - last[0]) * 1e3)
        last[0] = now

    lossy0 = ro.lossy_batches
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    last[0] = t0
    rec = run_episodes(ro, list(range(per_slot * E)), policy=None, max_steps=steps, settle_steps=settle, on_step=on_step)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    t1 = time.perf_counter()
    sink.close()
    drain = time.perf_counter() - t1
    g = gdev[: min(cnt[0], gdev.shape[0])].cpu().numpy()
    out = {"config": config, "envs": E, "episodes": int(rec.shape[0]), "steps_per_episode": steps, "settle_steps": settle, "env_steps": int(E * cnt[0]),
           "sustained_env_steps_per_s": E * cnt[0] / wall, "wall_s": wall, "step_latency_ms": latency_stats(lat[1:]),
           "slowest_steps": sorted(((round(m, 2), k) for k, m in enumerate(lat)), reverse=True)[:5],
           "re_rendered_batches": int(ro.lossy_batches - lossy0), "sink": {"frames_written": sink.frames_written, "steps_written": sink.steps_written,
                                                                          "producer_stalls": sink.stalls, "drain_after_run_s": drain, "format": sink.ext},
           "grasped_envs_max": int(g[:, 2].max()) if len(g) else 0, "grasped_env_steps_share_first_episode": float(g[:, 2].sum() / max(1, E * len(g))) if len(g) else 0.0,
           "mesh_contacts_max": int(g[:, 1].max()) if len(g) else 0, "candidates_max": int(g[:, 0].max()) if len(g) else 0,
           "kernel_flavours_sampled": sorted(flav), "records": summarize(rec),
           "note": "run_episodes on every slot: randomised per-slot resets (episode id -> grid pose), 30 settling steps + the episode's steps of the synthetic action "
                   "trace (closing ramp -> grasp -> lift 60 steps -> hold), get_obs() before every step (closed loop), sink on (every frame + state written by worker "
                   "processes); step_latency_ms = host wall clock between consecutive steps (the first step, which pays the first render, is left out)"}
    del ro
    if not sink_dir:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


class StubRollout:
    """CPU stand-in for BatchedRollout: the launcher / barrier / MAX / all-gather / JSON path of this file without a GPU
    (tests/test_distributed_gloo.py drives `bench.py --stub --gpus 2` through the same self-launch as the real bench)."""

    def __init__(self, n_env, rank):
        self.n_env, self.rank, self.t = n_env, rank, 0

    def step(self):
        time.sleep(0.002 * (1 + self.rank))
        self.t += 1


def run_stub(args, rank, world):
    import torch
    import torch.distributed as dist
    from r2s_hip import dist as rdist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tc0 = time.perf_counter()
    ro = StubRollout(args.envs or 4, rank)
    time.sleep(0.01 * (rank % 3))                      # ranks do not finish construction together
    construct_s = time.perf_counter() - tc0
    for _ in range(args.warmup):
        ro.step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ro.step()
    if world > 1:
        dist.barrier()
    elapsed = rdist.max_over_ranks(time.perf_counter() - t0, "cpu")
    records = rdist.gather_records([ro.n_env, args.steps, elapsed * 1e3, float(rank), 0.0, construct_s], "cpu")
    if rank == 0:
        print(json.dumps({"metric": "stub env-steps/s", "value": rdist.throughput(records, elapsed), "unit": "env-steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "data": "stub", "ranks_seen": int(records.shape[0]), "envs_total": int(records[:, 0].sum().item()),
                          "construct_s_per_rank": [round(float(x), 4) for x in records[:, 5]]}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="sloth_32env")
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default: the config's)")
    ap.add_argument("--substeps", type=int, default=667)
    ap.add_argument("--schedule", default=None, help="grasp | lissajous (default: the scene's; see r2s_hip/rollout.py)")
    ap.add_argument("--res", default=None, help="WxH frame size instead of the config's (the reference's default frame: 848x480, cfg/env/xarm_gripper.yaml:21-49)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the throughput-mode comparison that follows the timed window")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--stub", action="store_true", help="CPU stand-in workload over gloo (launcher test)")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the parity gate that precedes the timed window (default: on; a failing gate "
                                                                  "withholds `value`)")
    ap.add_argument("--dephase", type=int, default=-1, help="the timed window's environments run the action trace (e K) // E steps late — they close their "
                                                            "grippers one after the other across the window instead of all at its middle (episodes of "
                                                            "eval_policy_parallel.py do not share a phase).  -1: K = --steps (default), 0: all environments "
                                                            "in phase (the round-4 window; then no second, synchronised window is timed)")
    ap.add_argument("--open-loop", action="store_true", help="the timed window only ENQUEUES steps (no get_obs() per step): the round-4 definition of `value`; "
                                                             "for profiling runs (a kernel trace of back-to-back steps), never for a reported number")
    ap.add_argument("--close-rate", type=float, default=0.1, help="gripper scenes: the commanded opening falls by this much per env step once the gripper closes "
                                                                     "(towards 0: where the fingers stop is the grasp state machine's decision, phystwin.py:399-405); 0: the "
                                                                     "command jumps to its closed value within one env step (rounds 1-5: no grasp can ever be detected)")
    ap.add_argument("--episodes", type=int, default=-1, help="full episodes per environment slot of the sustained measurement that follows the timed window "
                                                              "(reset -> 30 settling steps -> the scene's episode length, sink on; -1: 2 on one GPU, 0 on several; 0: skip)")
    ap.add_argument("--episode-steps", type=int, default=0, help="policy steps per episode of the sustained measurement (0: the reference's: sloth 450, rope 900, T 1800)")
    ap.add_argument("--sink", default=None, help="directory: also run the observation sink (row f4) every step — packed 8-bit frames + state "
                                                 "to pinned ring buffers, JPEG / pickle written by a host thread; not part of the headline")
    args = ap.parse_args()

    from r2s_hip import dist as rdist

    # --gpus N from a plain process: start N ranks (one per GPU) and exit with their code; a rank checks WORLD_SIZE == N
    rank, local_rank, world = rdist.resolve_world(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if args.stub:
        return run_stub(args, rank, world)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    from r2s_hip.rollout import CONFIGS, BatchedRollout

    res = tuple(int(v) for v in args.res.lower().split("x")) if args.res else None
    # ---- parity gate (every timed configuration first passes the position and image gates) ----------------
    # THIS workload through the product path into the phase the window times in contact, 20 substeps + the side and wrist frames
    # against the oracle (oracle/parity_gate.py: the oracle is the checker, never the thing timed).  A multi-environment window is
    # gated on a 9-environment batch — the large-batch layout with two concurrent kernel chains, the flavour family the window
    # times (the first and the last environment, one per chain, each against its own oracle) —, a one-environment window on one
    # environment (the resident layout).  `parity_gate.flavour / chains / layout` say what ran.  Rank 0 only; the other ranks wait
    # at the first barrier.
    gate = None
    if not args.no_parity_gate and rank == 0:
        from oracle import parity_gate
        try:
            n_bench = args.envs if args.envs is not None else CONFIGS[args.config][3]
            gate = parity_gate.run(args.config, device=dev, seed=rank, num_substeps=args.substeps, n_compare=20, close_at=2,
                                   n_env=9 if n_bench >= 9 else n_bench, res=res,
                                   close_rate=args.close_rate if args.close_rate > 0 and "pusher" not in args.config else None)
        except Exception as e:  # a gate that cannot run is a failed gate
            gate = {"passed": False, "error": f"{type(e).__name__}: {e}"}

    # Gripper scenes close at --close-rate per env step (a ramp of ~10 steps; the grasp latches ~8 steps after the closing starts): the
    # timed window begins with the first environments 4 steps into their closing and the delays cover HALF the window, so that every
    # environment is in the grasp — holding, lifting — for the window's last third.  The pusher scene has no ramp: the rod reaches the
    # block at the window's start + the environment's delay, delays over the whole window (as in round 5).
    ramp = args.close_rate > 0 and "pusher" not in args.config
    kw_ro = dict(device=dev, seed=rank, n_env=args.envs, num_substeps=args.substeps, schedule=args.schedule, res=res)
    if ramp:
        kw_ro.update(close_rate=args.close_rate, lift_steps=60)
    lead = 4 if ramp else 0                      # env steps of closing before the window starts
    close_sync = args.warmup + (max(0, args.steps // 2 - 6) if ramp else args.steps // 2)   # all environments in phase: free motion, then closing -> contact -> grasp -> lift
    K_dephase = (max(2, args.steps // 2) if ramp else args.steps) if args.dephase < 0 else args.dephase
    tc0 = time.perf_counter()
    ro = BatchedRollout(args.config, close_at=max(0, args.warmup - lead) if K_dephase > 1 else close_sync, **kw_ro)
    dephased = K_dephase > 1 and ro.with_gripper and ro.schedule in ("grasp", "push") and ro.n_env > 1   # (one environment: its contact event at the window's middle)
    if dephased:
        ro.set_dephase(K_dephase)
    elif K_dephase > 1:   # a trace without a contact event (the small test scenes' lissajous path), or one environment: nothing to de-phase
        ro = BatchedRollout(args.config, close_at=close_sync, **kw_ro)

    torch.cuda.synchronize(dev)
    construct_s = time.perf_counter() - tc0            # scene synthesis, topology upload, graph capture of every flavour, settling
    closed_loop = not args.open_loop

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    sink = None
    if args.sink:
        from r2s_hip.sink import ObservationSink
        sink = ObservationSink(os.path.join(args.sink, f"rank{rank}"), ro.n_env, ro.views, ro.H, ro.W, device=dev, slots=6,
                               state_bytes=2 * ro.n_env * ro.N * 12 + 4096)
    for _ in range(args.warmup):
        ro.step()
        if closed_loop:
            ro.get_obs()
    barrier()
    # per-step stamps on the launch stream (torch's current stream is the one every kernel of the step is enqueued on):
    # step boundaries + the physics graph alone.  Contact counters are logged on the device, read after the window.
    ro.start_log(args.steps)
    lossy_before = ro.lossy_batches
    t0 = time.perf_counter()
    lat = []
    for k in range(args.steps):
        ts = time.perf_counter()
        ro.step()
        if closed_loop:
            ro.get_obs()          # the policy's read: wait for the frames of THIS step, validate the batch, re-render a lossy one
        if sink is not None:
            sink.submit(k, ro.out_color, state=dict(x=ro.phys.x, v=ro.phys.v), ready=getattr(ro, "_render_done", None))
        lat.append((time.perf_counter() - ts) * 1e3)   # (closed loop: action -> observation, host wall clock; open loop: enqueue time only)
    barrier()
    elapsed = time.perf_counter() - t0
    re_rendered = int(ro.lossy_batches - lossy_before)
    log = ro.read_log()
    ro._poll_raster(wait=True)   # the sync-free raster pipeline: the count of the LAST batch, and how many batches of the window overflowed
    sink_report = None
    if sink is not None:
        t1 = time.perf_counter()
        sink.close()
        sink_report = {"frames_written": sink.frames_written, "steps_written": sink.steps_written, "producer_stalls": sink.stalls,
                       "drain_after_window_s": time.perf_counter() - t1, "format": sink.ext,
                       "note": "submit() only enqueues a pack kernel + one async D2H per step; encoding and file writes run in worker processes"}
    # N > 1: slowest rank's time (MAX) and the metric all-gather of north_star — one fixed-size record per rank
    elapsed = rdist.max_over_ranks(elapsed, dev)
    n_success = int(ro.success_flags().sum().item())  # device-side task predicate (row f4); outside the timed region
    records = rdist.gather_records([ro.n_env, args.steps, elapsed * 1e3, float(ro.last_num_rendered), float(n_success), construct_s], dev)
    total_envs = int(records[:, 0].sum().item())

    # ---- the synchronised window (outside the timed region, reported next to `value`, never as `value`): the same K closed-loop steps with
    # all environments in phase — free motion first half, every gripper closing at its middle, contact second half (`phases` splits it; round 4
    # timed this one, enqueue-only, as `value`) — on a rollout of its own; the accounting runs below continue from the state it ends in
    sync_report, plog, ra = None, log, ro
    if dephased and not args.stub:
        ra = BatchedRollout(args.config, close_at=close_sync, **kw_ro)
        for _ in range(args.warmup):
            ra.step()
            if closed_loop:
                ra.get_obs()
        torch.cuda.synchronize(dev)
        ra.start_log(args.steps)
        t0s = time.perf_counter()
        for _ in range(args.steps):
            ra.step()
            if closed_loop:
                ra.get_obs()
        torch.cuda.synchronize(dev)
        els = time.perf_counter() - t0s
        plog = ra.read_log()
        # ... and once more as round 4 timed its `value`: the same synchronised schedule, steps only ENQUEUED (comparable with BENCH_r04.json)
        rb = BatchedRollout(args.config, close_at=close_sync, **kw_ro)
        for _ in range(args.warmup):
            rb.step()
        torch.cuda.synchronize(dev)
        t0e = time.perf_counter()
        for _ in range(args.steps):
            rb.step()
        torch.cuda.synchronize(dev)
        enq_rate = rb.n_env * args.steps / (time.perf_counter() - t0e)
        del rb
        sync_report = {"env_steps_per_s": ra.n_env * args.steps / els, "ms_per_step": els / args.steps * 1e3, "closed_loop": closed_loop,
                       "enqueue_only_env_steps_per_s": enq_rate,
                       "note_enqueue_only": "the round-4 definition of `value` (synchronised schedule, steps enqueued back to back, no get_obs()): for comparison with BENCH_r04.json only",
                       "note": "this rank; the same workload with all environments in phase (every gripper closes at the window's middle): the window's "
                               "first half runs the free flavour, its second half the contact flavour — see `phases`"}

    # throughput mode (outside the timed region): the rollout continues from the state the window ended in, first serially, then with the
    # rasterisation of step t on a second stream next to the substeps of step t+1 (BatchedRollout.set_pipelined) — what an open-loop stretch
    # (an action chunk) can run at
    pipe_report = None
    if not args.stub and not args.no_pipelined:
        kp = max(4, min(args.steps, 10))
        rates = []
        for mode in (False, True):
            ra.set_pipelined(mode)
            for _ in range(2):
                ra.step()
            ra.wait_render()
            torch.cuda.synchronize(dev)
            t0p = time.perf_counter()
            for _ in range(kp):
                ra.step()
            ra.wait_render()
            torch.cuda.synchronize(dev)
            rates.append(ra.n_env * kp / (time.perf_counter() - t0p))
        ra.set_pipelined(False)
        pipe_report = {"serial_env_steps_per_s": rates[0], "pipelined_env_steps_per_s": rates[1], "steps": kp,
                       "note": "this rank, after the synchronised window, in the state it ended in (contact), ENQUEUE-ONLY: the same steps serially and with the "
                               "rasterisation of env step t on a second stream next to the substeps of step t+1; results are bit-identical (tested).  Valid when the "
                               "next action does not depend on this step's observation (inside an action chunk); never what `value` reports"}

    # closed-loop accounting (outside the timed region): the same state, the same number of steps, first enqueue-only, then with get_obs() after
    # every step — the ratio is what the policy's read costs (`value` is measured WITH it)
    closed_report = None
    if not args.stub:
        kc = max(4, min(args.steps, 10))
        rates = []
        lossy0 = 0
        for with_obs in (False, True):
            for _ in range(2):
                ra.step()
            ra.get_obs()
            torch.cuda.synchronize(dev)
            lossy0 = ra.lossy_batches
            t0c = time.perf_counter()
            for _ in range(kc):
                ra.step()
                if with_obs:
                    ra.get_obs()
            torch.cuda.synchronize(dev)
            rates.append(ra.n_env * kc / (time.perf_counter() - t0c))
        closed_report = {"env_steps_per_s": rates[1], "enqueue_only_env_steps_per_s": rates[0], "ratio": rates[1] / rates[0], "steps": kc,
                         "re_rendered_batches": int(ra.lossy_batches - lossy0),
                         "note": "this rank, after the synchronised window, in the state it ended in (contact): the same steps enqueue-only and with "
                                 "get_obs() after every step (host waits for the frame, validates the sync-free raster batch, re-renders a lossy one)"}
    if ra is not ro:
        del ra

    # cross-check for the roofline (untimed): the same env step captured as ONE kernel per batched substep, so that the
    # HIP-event time / 667 is a per-kernel duration that rocprofv3's per-kernel average can be compared with directly
    single_us = None
    ro.phys.set_timing(True)
    if ro.phys.layout_stats()["chains"] > 1:
        ro.phys.set_tuning(chains=1)          # drops the captured graphs; the next steps re-capture with one chain
        for _ in range(3):
            ro.physics_step()
        torch.cuda.synchronize(dev)
        ms1, k1 = ro.phys.last_step_ms()
        single_us = ms1 / max(k1, 1) * 1e3
        ro.phys.set_tuning(chains=0)
    # small batches: the same env step with the per-substep kernels of the same layout instead of the resident launch (untimed)
    per_substep_us = None
    if ro.phys.last_flavour().get("resident"):
        ro.phys.set_resident(False)
        best = None
        for _ in range(6):                        # the first steps capture their graph flavour (both parities of the state buffer): best of six
            ro.physics_step()
            torch.cuda.synchronize(dev)
            ms1, k1 = ro.phys.last_step_ms()
            best = ms1 / max(k1, 1) * 1e3 if best is None else min(best, ms1 / max(k1, 1) * 1e3)
        per_substep_us = {"avg_launch_us": best, "kernel": ro.phys.last_flavour()["kernel"],
                          "note": "r2s_phys_set_resident(0): one launch per substep (large batches' flavour rule: deferred queries while anything is "
                                  "within 3 cm of a mesh), best of six env steps after the timed region"}
        ro.phys.set_resident(True)
    ro.phys.set_timing(False)

    # stage timing of the raster pipeline (separate, untimed pass; drained first: the candidate rebuild of the last physics step runs
    # on its own stream and would otherwise sit inside the stage events)
    torch.cuda.synchronize(dev)
    ro.raster.set_timing(True)
    for _ in range(2):                      # the first timed render is the first synchronous one of the run (scratch re-sized on the host
        ro.render()                         # between two stage events): the second is the one reported
        torch.cuda.synchronize(dev)
    stages = ro.raster.stage_ms()
    ro.raster.set_timing(False)
    frames = ro.n_env * ro.views
    raster_ms = sum(stages.values())
    # realised scene statistics: visible Gaussians, instances, tile list lengths
    dbg = ro.raster.debug()
    tiles = ((ro.W + 15) // 16) * ((ro.H + 15) // 16)
    from r2s_hip.raster import _memcpy_d2d
    rng_t = torch.empty(frames * tiles, 2, dtype=torch.int32, device=dev)
    _memcpy_d2d(rng_t.data_ptr(), dbg["ranges_ptr"], rng_t.numel() * 4, dev)
    lens = (rng_t[:, 1] - rng_t[:, 0]).float()
    scene = {"gaussians_per_frame": ro.P, "visible_fraction": float((dbg["radii"] > 0).float().mean()), "instances": int(dbg["num_rendered"]),
             "tile_list_mean": float(lens.mean()), "tile_list_max": int(lens.max())}
    # skinning (row f1): torch events are valid here, the kernels run on torch's current stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ro._update_means()
    e1.record()
    torch.cuda.synchronize(dev)
    skin_ms = e0.elapsed_time(e1) / 5

    if rank == 0:
        value = total_envs * args.steps / elapsed
        n_sub = args.substeps
        phys_ms = sum(log["phys_ms"]) / args.steps                 # mean physics graph time per env step over the timed window
        t_kernel = phys_ms * 1e-3 / n_sub
        alg_bytes = ro.physics_algorithmic_bytes_per_substep()
        chains = ro.phys.layout_stats()["chains"]
        achieved = alg_bytes / t_kernel / 1e9
        comp_bytes = ro.composite_algorithmic_bytes()
        comp_gbs = comp_bytes / (stages["composite"] * 1e-3) / 1e9 if stages["composite"] > 0 else 0.0

        def phase(sel):   # of the SYNCHRONISED window (the timed one itself when it was not de-phased)
            idx = [i for i in range(args.steps) if sel(i)]
            if not idx:
                return None
            return {"steps": len(idx), "ms_per_step": sum(plog["step_ms"][i] for i in idx) / len(idx),
                    "physics_ms_per_step": sum(plog["phys_ms"][i] for i in idx) / len(idx),
                    "substep_us": sum(plog["phys_ms"][i] for i in idx) / len(idx) / n_sub * 1e3,
                    "self_collision_candidates": int(max(plog["candidates"][i] for i in idx)),
                    "mesh_contacts": int(max(plog["mesh_hits"][i] for i in idx)),
                    "grasped_envs": int(max(plog["grasped"][i] for i in idx)),
                    "kernel_flavours": sorted({plog["flavour"][i] for i in idx})}

        resident_steps = sum("k_steps_resident" in f for f in log["flavour"])   # env steps of the window that ran as one resident launch
        first_contact = (close_sync if dephased else ro.close_at) - args.warmup + (1 if ramp else 0)  # index in the synchronised window of the step in which the pads reach the object (the ramp's second step: the first particles enter the pads' margins) / the fingers snap shut / the rod arrives
        (pmc_sub, src), (pmc_comp, _) = pmc_summary("k_substep", args.config), pmc_summary("k_composite", args.config)
        shared_bytes = 16 * ro.S + 48 * ro.N * ro.n_env        # the topology once (it is shared by the environments and L2-resident) + every environment's state
        traffic = pmc_sub["hbm_bytes_per_launch"] if pmc_sub and ro.n_env == 32 and n_sub == 667 else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_shared_topology": shared_bytes / t_kernel / 1e9 / HBM_PEAK_GBS,
                "shared_topology_bytes_per_launch": shared_bytes,
                "traffic_over_shared": (traffic / shared_bytes) if traffic else None,
                "bound_in_practice": "VALU issue + dependent latency (valu_busy_frac of the SIMD issue cycles; hbm_actual_frac of the HBM peak): neither "
                                     "pipe is saturated — `frac` is the SURVEY.md §8d CONTRACT figure (every environment charged its own copy of the "
                                     "topology), `frac_shared_topology` charges the topology once and is the physical lower bound on HBM bytes",
                "traffic": traffic,
                "traffic_source": src if pmc_sub else None,
                "hbm_actual_frac": (pmc_sub["hbm_bytes_per_launch"] / t_kernel / 1e9 / HBM_PEAK_GBS) if pmc_sub and ro.n_env == 32 and n_sub == 667 else None,
                "valu_busy_frac": pmc_sub.get("valu_busy_frac") if pmc_sub else None,
                # the kernel's own arithmetic bound: its VALU instructions x 4 issue cycles over the chip's 1024 SIMDs, as a share of the
                # kernel's span — the same ratio as valu_busy_frac, named for what it says: 1 / valu_floor_frac is how far the kernel runs above its VALU floor
                "valu_floor_frac": pmc_sub.get("valu_busy_frac") if pmc_sub else None,
                "valu_floor_us": (pmc_sub["counters_mean_per_dispatch"]["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e3) if pmc_sub and "SQ_INSTS_VALU" in pmc_sub.get("counters_mean_per_dispatch", {}) else None,
                "read_requests_reaching_dram_frac": pmc_sub.get("read_requests_reaching_dram_frac") if pmc_sub else None,
                "counters_stale": pmc_sub.get("stale") if pmc_sub else None,
                "kernel": ("k_steps_resident (small batch: all substeps of an env step in ONE launch; the figures below are per substep of it)" if resident_steps == args.steps
                           else "k_substep (fused spring gather + velocity + collisions + integrate)"),
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": t_kernel * 1e6, "launches": n_sub * args.steps,
                "resident_env_steps": resident_steps, "per_substep_kernels_check": per_substep_us,
                "concurrent_chains": chains,
                "single_chain_check": None if single_us is None else {
                    "avg_launch_us": single_us, "achieved": alg_bytes / (single_us * 1e-6) / 1e9, "frac": alg_bytes / (single_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "note": "same step with one kernel per batched substep (r2s_phys_set_tuning chains=1, measured after the timed region, "
                            "in the contact state the window ended in): a per-kernel duration, comparable with profiles/r3_bench_kernel_stats_chains1.md (that table is of the free + contact window, this figure of the contact state alone)"},
                "note": f"frac is the SURVEY.md §8d contract figure: algorithmic bytes (16 S + 48 N per env) / avg launch time / 8 TB/s, averaged over "
                        f"the whole timed window (free + contact).  The topology (16 S) is shared by the {ro.n_env} envs and stays in L2, so the bytes that "
                        "reach HBM are hbm_actual_frac of peak; valu_busy_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel span), the share of SIMD issue cycles spent on VALU instructions: the kernel is "
                        f"VALU-issue / latency bound, not HBM bound.  One 'launch' = one batched substep of all {ro.n_env} envs, issued as {chains} concurrent "
                        "kernels over disjoint env ranges, each chain its own captured graph on its own stream (per-kernel durations overlap); avg_launch_us = HIP-event time of the 667-substep step / 667"}
        if roof["frac"] > 1.0:
            roof["frac_exceeds_one"] = ("the contract charges 16 S bytes of topology PER ENVIRONMENT; the environments share one copy that stays in L2, so "
                                        "the contract figure is not bounded by 1 — read frac_shared_topology / hbm_actual_frac for the physical picture")
        out = {
            "metric": "sim env-steps/sec (phys+render) per node at 32 envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {ro.N} particles / {ro.S} springs / {ro.P} Gaussians per env, "
                                   f"{ro.n_env} envs per GPU, {args.substeps} substeps + {ro.views} frames {ro.W}x{ro.H} per env step; "
                                   f"action trace '{ro.schedule}'" + (f", environment e {('closes its gripper' if ro.schedule == 'grasp' else 'reaches the block')} at timed step (e * {K_dephase}) // {ro.n_env} "
                                                                        f"(de-phased episodes); " if dephased else f": free motion, contact from timed step {first_contact}; ")
                                   + ("every step ends in get_obs() (closed loop)" if closed_loop else "steps only enqueued (--open-loop)"),
                       "envs_per_gpu": ro.n_env, "parallelism": f"envs sharded over {world} GPU(s), no data-path collective"},
            "window": {"closed_loop_get_obs_every_step": closed_loop, "dephased": bool(dephased), "K": int(K_dephase) if dephased else 0,
                       "re_rendered_batches": re_rendered, "step_latency_ms": dict(latency_stats(lat), per_step=[round(m, 3) for m in lat]),
                       "close_rate": args.close_rate if ramp else None,
                       "substep_us_per_step": [round(m / n_sub * 1e3, 2) for m in log["phys_ms"]], "mesh_contacts_per_step": log["mesh_hits"],
                       "grasped_envs_per_step": log["grasped"], "kernel_flavours": sorted(set(log["flavour"])),
                       "note": "what `value` was timed over: `steps` steps, each followed by get_obs() (the host waits for the step's frames, validates the sync-free "
                               "raster batch, re-renders a lossy one) before the next step is applied; environment e runs the action trace (e K) // E steps late.  Gripper "
                               "scenes: the commanded opening ramps down by close_rate per step from 4 steps before the window on, the pads load up, the grasp state machine "
                               "latches from the stepper's own forces (grasped_envs_per_step) and the object is lifted in the grasp; step_latency_ms = host wall clock per step"},
            "phases": {"free": phase(lambda i: i < first_contact), "contact": phase(lambda i: i >= first_contact),
                       "window": "synchronised_window" if dephased else "the timed window",
                       "note": "free = the end effector moves, nothing touches; contact = fingers closed on the toy's arms / rod against the block "
                               "(mesh_contacts = particles inside a collision margin in the last substep, self_collision_candidates = particles "
                               "with live candidates, maxima over the phase's steps)"},
            "parity_gate": gate if gate is not None else {"passed": None, "note": "skipped (--no-parity-gate)"},
            "roofline": roof,
            "raster": {"gs_raster_mpix_per_s": frames * ro.W * ro.H / (raster_ms * 1e-3) / 1e6, "frames": frames,
                       "num_rendered": int(ro.last_num_rendered), "stage_ms": stages, "scene": scene,
                       "lossy_batches": int(ro.lossy_batches),   # sync-free batches whose instance count outgrew the capacity (must be 0)
                       "composite_roofline": {"bound": "hbm", "achieved": comp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": comp_gbs / HBM_PEAK_GBS, "algorithmic_bytes": comp_bytes,
                                              "traffic": pmc_comp["hbm_bytes_per_launch"] if pmc_comp and ro.n_env == 32 else None,
                                              "traffic_source": src if pmc_comp else None,
                                              "valu_busy_frac": pmc_comp.get("valu_busy_frac") if pmc_comp else None,
                                              "note": "VALU/exp-bound (≈115 flop per algorithmic byte, SURVEY.md §7): the north_star's >= 0.60 of HBM peak "
                                                      "is not reachable for this kernel at any instruction count above ~1/3 of the reference's per-pixel "
                                                      "arithmetic; valu_busy_frac (SQ_INSTS_VALU x 4 cycles over the SIMD cycles of the kernel span, at most 1) is the figure that says how close to its real bound it runs"}},
            "physics_ms_per_env_step": phys_ms, "skinning_ms_per_env_step": skin_ms,
            "construct_s_per_rank": [round(float(x), 3) for x in records[:, 5]],   # outside the timed region; every rank builds, captures and settles its own batch
            "task_success": {"envs_satisfying_predicate": int(records[:, 4].sum().item()), "of": total_envs,
                             "note": "frame-level success predicate of the scene's task evaluated on the device after the last step "
                                     "(synthetic action trace: not a policy result)"},
        }
        if sink_report is not None:
            out["observation_sink"] = sink_report
        if pipe_report is not None:
            out["throughput_mode"] = pipe_report
        if closed_report is not None:
            closed_report["ratio_to_value"] = closed_report["env_steps_per_s"] / (ro.n_env * args.steps / elapsed) if world == 1 else None
            out["closed_loop_get_obs"] = closed_report
        if ro.lossy_batches:
            out["lossy_batches_in_run"] = int(ro.lossy_batches)
            out["note_lossy"] = ("sync-free raster batches overflowed their capacity during this run; get_obs() rendered those steps again before handing "
                                 "the frames out (window.re_rendered_batches of them inside the timed window: their cost is in `value`)")
        if sync_report is not None:
            sync_report["value_over_this"] = (ro.n_env * args.steps / elapsed) / sync_report["env_steps_per_s"] if world == 1 else None
            out["synchronised_window"] = sync_report
        if gate is not None and not gate.get("passed"):
            out["value_withheld"] = out["value"]
            out["value"] = None
            out["note"] = "PARITY GATE FAILED: the throughput of a path whose results differ from the reference's is not a result (see parity_gate)"
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(ro, args.cpu_budget)
            except Exception as e:  # the baseline is a report, never the product
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        n_epi = args.episodes if args.episodes >= 0 else (2 if world == 1 else 0)
        if n_epi > 0 and not args.stub:
            ro = None
            torch.cuda.empty_cache()
            epi = []
            for cfg_e, per_slot in ((args.config, n_epi),) + ((("T_pusher_32env", 1),) if args.config == "sloth_32env" and args.episode_steps == 0 else ()):
                try:
                    epi.append(sustained_episodes(cfg_e, dev, rank, args.envs if cfg_e == args.config else None, args.substeps, res if cfg_e == args.config else None,
                                                  per_slot, args.episode_steps, args.close_rate, args.sink))
                except Exception as e:   # reported, never fatal for the line
                    epi.append({"config": cfg_e, "error": f"{type(e).__name__}: {e}"})
            out["episodes"] = {"runs": epi, "sustained_over_value": (epi[0].get("sustained_env_steps_per_s") or 0.0) / value if epi and value else None,
                               "note": "sustained full-episode figures NEXT to the timed window (never `value`); the first run is this config's"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
