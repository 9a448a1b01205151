cd $GRAFT_REPO_ROOT
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 7 2>&1 | tail -7 | cut -c1-110
R2S_NO_BANK_ORDER=1 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 2 2>&1 | tail -2 | cut -c1-110
timeout 600 python -m pytest tests -m gpu -x -q -k "physics or eef or dynamics or full_size or randomized" 2>&1 | tail -3
