cd $GRAFT_REPO_ROOT
for n in old b14s0 old b14s0; do echo $n; R2S_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libr2s_$n.so timeout 200 python tools/profiling/raster_bench.py sloth_32env 2>&1 | tail -1 | cut -c1-200; done
