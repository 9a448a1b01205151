"""World-size-2 gloo test of the N>1 path of bench.py: env sharding + MAX time + all-gather of rank records."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "real2sim-eval_amd"))
    from r2s_hip import dist as rdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rdist.env_shard(65, rank, world)           # uneven split on purpose
    elapsed = 1.0 + rank                                 # rank 1 is slower
    tmax = rdist.max_over_ranks(elapsed, "cpu")
    rec = rdist.gather_records([hi - lo, 3, elapsed * 1e3, 7.0 + rank], "cpu")
    q.put((rank, lo, hi, tmax, rec.tolist(), rdist.throughput(rec, tmax)))
    dist.barrier()
    dist.destroy_process_group()


def test_env_shard_is_a_contiguous_partition():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real2sim-eval_amd"))
    from r2s_hip.dist import env_shard

    for n, w in ((256, 8), (65, 2), (7, 8), (32, 1)):
        spans = [env_shard(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert env_shard(256, 3, 8) == (96, 128)  # config C3: 256 envs over 8 GPUs, 32 each


@pytest.mark.timeout(120)
def test_two_rank_gloo_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, rec0, thr0), (r1, lo1, hi1, t1, rec1, thr1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 33, 33, 65)
    assert t0 == t1 == 2.0                      # MAX over ranks
    assert rec0 == rec1 and rec0[0][:2] == [33.0, 3.0] and rec0[1][:2] == [32.0, 3.0]
    assert thr0 == thr1 == pytest.approx((33 * 3 + 32 * 3) / 2.0)


def _run_bench(args, env=None, timeout=240):
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if env is None:
            e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=e, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


@pytest.mark.timeout(300)
def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` from a plain process must start 2 ranks (r2s_hip.dist.self_launch -> torch.distributed.run)
    and report n_gpus == 2 — the reference's eval_policy_parallel.py:266-280 spawns its workers the same way.  `--stub`
    swaps the GPU rollout for a sleep so that the launcher, barrier, MAX and all-gather run here over gloo."""
    rc, out, err = _run_bench(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--envs", "5"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["envs_total"] == 10 and out["steps"] == 3
    # whole-job throughput = envs of both ranks * steps / slowest rank's time (rank 1 sleeps twice as long per step)
    assert out["value"] == pytest.approx(10 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)
    assert out["ms_per_step"] >= 4.0


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    rc, out, err = _run_bench(["--gpus", "2", "--stub", "--steps", "1", "--warmup", "0"], env=env)
    assert rc == 2 and out is None and "WORLD_SIZE=1" in err


def test_launch_command_shape():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real2sim-eval_amd"))
    from r2s_hip.dist import launch_command

    cmd = launch_command(8, "bench.py", ["--gpus", 8, "--config", "T_pusher_32env"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["bench.py", "--gpus", "8", "--config", "T_pusher_32env"]


@pytest.mark.timeout(600)
def test_bench_gpus_8_stub_dry_run_of_the_round_end_scaling_launch():
    """The driver's N = 8 launch, dry: `bench.py --gpus 8 --stub` self-launches eight ranks over gloo — rendezvous on 127.0.0.1,
    barrier, MAX over ranks, the all-gather of one record per rank — and reports every rank's construction time, so that the
    first real 8-GPU run has nothing left to discover but the number."""
    rc, out, err = _run_bench(["--gpus", "8", "--stub", "--steps", "3", "--warmup", "1", "--envs", "32"], timeout=500)
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["envs_total"] == 256 and out["scaling"] == "weak"
    assert len(out["construct_s_per_rank"]) == 8 and all(c >= 0 for c in out["construct_s_per_rank"])
    # the slowest rank (rank 7 sleeps 8 x 2 ms per step) sets the time; whole-job value = 256 envs x steps / that time
    assert out["value"] == pytest.approx(256 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6) and out["ms_per_step"] >= 16.0
