set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1/pytest.log
tail -15 gpurun_out/c1/pytest.log
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 4 6 14 > gpurun_out/c1/diag.log 2>&1; tail -20 gpurun_out/c1/diag.log
timeout 400 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; tail -c 3000 gpurun_out/c1/bench.json; tail -5 gpurun_out/c1/bench.err
timeout 300 python bench.py --config T_pusher_32env --steps 20 --warmup 10 --no-cpu-baseline > gpurun_out/c1/bench_pusher.json 2> gpurun_out/c1/bench_pusher.err; tail -c 1500 gpurun_out/c1/bench_pusher.json
