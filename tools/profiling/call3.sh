cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c3; mkdir -p $out
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 6 12 > $out/diag32.log 2>&1; tail -13 $out/diag32.log
timeout 300 python tools/profiling/grasp_diag.py T_pusher_32env 32 4 10 > $out/diagp.log 2>&1; tail -11 $out/diagp.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/tools/profiling/grasp_diag.py sloth_32env 32 3 7 > $out/trace.log 2>&1 || echo trace-failed
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $out/kernel_stats.md > /dev/null 2>&1 || echo stats-failed
head -24 $out/kernel_stats.md | cut -c1-200
rm -rf $out/trace
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -25 $out/pytest.log
