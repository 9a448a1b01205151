cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c9; mkdir -p $out
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 7 > $out/diag32.log 2>&1; tail -7 $out/diag32.log | cut -c1-110
