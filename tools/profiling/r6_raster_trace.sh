# kernel trace of the raster path alone (64 frames of sloth_32env, 6 batches): per-kernel table -> gpurun_out/r6_raster_kernels.md
cd /root/repo
out=gpurun_out/raster_trace; rm -rf $out; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out -o rb -- python $GRAFT_REPO_ROOT/tools/profiling/raster_bench.py ${1:-sloth_32env} > $GRAFT_REPO_ROOT/$out/run.log 2>&1 )
tail -1 $out/run.log
db=$(find $out -name "*_results.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/r6_raster_kernels.md | grep -v "k_substep\|k_skin\|k_bone\|resting\|candidates\|k_tri_pre\|k_zero\|rocclr" | head -40
rm -rf $out
