cd $GRAFT_REPO_ROOT
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 7 2>&1 | tail -8 | cut -c1-110
R2S_RCAP=1024 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 2 2>&1 | tail -2 | cut -c1-110
R2S_CHAINS=1 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 6 2>&1 | tail -6 | cut -c1-110
