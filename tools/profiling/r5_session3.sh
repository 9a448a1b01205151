#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s3; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
export R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so
for ch in 1 4; do
  R2S_CHAINS=$ch timeout 200 python tools/probes/pf_probe.py sloth_32env 32 2 6 > $out/pf_probe_sloth_c$ch.txt 2>&1; tail -12 $out/pf_probe_sloth_c$ch.txt
done
R2S_CHAINS=1 timeout 200 python tools/probes/pf_probe.py T_pusher_32env 32 2 6 > $out/pf_probe_pusher_c1.txt 2>&1; tail -12 $out/pf_probe_pusher_c1.txt
