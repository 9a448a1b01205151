cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c10; mkdir -p $out
R2S_CHAINS=1 R2S_HIP_LIB=$R/scratch/libr2s_probe.so TAG=r2a timeout 300 python tools/probes/phase_probe.py > $out/probe.log 2>&1; tail -12 $out/probe.log
