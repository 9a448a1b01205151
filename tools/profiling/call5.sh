cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c5; mkdir -p $out
for ch in 2 3 4; do
R2S_CHAINS=$ch timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 10 > $out/diag32_c$ch.log 2>&1; echo "chains $ch"; tail -10 $out/diag32_c$ch.log | cut -c1-120
done
timeout 300 python tools/profiling/grasp_diag.py T_pusher_32env 32 4 9 > $out/diagp.log 2>&1; tail -9 $out/diagp.log | cut -c1-130
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -25 $out/pytest.log
