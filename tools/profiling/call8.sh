cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c8; mkdir -p $out
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 8 > $out/diag32.log 2>&1; tail -8 $out/diag32.log | cut -c1-110
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
