# per-queue timeline of the substep kernels of the headline step (four chains), free flight and held grasp
cd /root/repo
out=gpurun_out/chain_tl; rm -rf $out; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out -o tl -- python $GRAFT_REPO_ROOT/tools/profiling/grasp_diag.py sloth_32env 32 3 ${1:-14} 0.1 > $GRAFT_REPO_ROOT/$out/run.log 2>&1 )
grep "step  2:\|step 13" $out/run.log | cut -c1-110
db=$(find $out -name "*_results.db" | head -1)
python tools/profiling/chain_timeline.py $db "k_substep<" | tee gpurun_out/r6_chain_timeline.txt
python tools/profiling/chain_timeline.py $db "k_substep_pf<256, 1024, true" | tee -a gpurun_out/r6_chain_timeline.txt
rm -rf $out
