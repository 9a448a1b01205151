#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s8; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 900 python tools/profiling/variant_bench.py tri_sub:default no_tri_sub:default:R2S_NO_TRI_SUB=1 two_launch:default:R2S_PF=0 > $out/variant_sloth.txt 2>&1; tail -4 $out/variant_sloth.txt
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py sloth_32env 32 2 6 > $out/pf_probe_sloth_c4.txt 2>&1; tail -9 $out/pf_probe_sloth_c4.txt
VB_CONFIG=rope_1env timeout 300 python tools/profiling/variant_bench.py tri_sub:default no_tri_sub:default:R2S_NO_TRI_SUB=1 > $out/variant_rope.txt 2>&1; tail -3 $out/variant_rope.txt
bash tools/profiling/r5_gputests.sh
