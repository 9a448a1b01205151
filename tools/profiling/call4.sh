cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c4; mkdir -p $out
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 6 14 > $out/diag32.log 2>&1; tail -15 $out/diag32.log
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -25 $out/pytest.log
