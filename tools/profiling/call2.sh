cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 4 6 14 > gpurun_out/c2/diag.log 2>&1; tail -16 gpurun_out/c2/diag.log
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 6 12 > gpurun_out/c2/diag32.log 2>&1; tail -13 gpurun_out/c2/diag32.log
timeout 300 python tools/profiling/grasp_diag.py T_pusher_32env 32 4 10 > gpurun_out/c2/diagp.log 2>&1; tail -11 gpurun_out/c2/diagp.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c2/pytest.log
tail -25 gpurun_out/c2/pytest.log
