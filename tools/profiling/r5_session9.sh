#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s9; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 900 python tools/profiling/variant_bench.py per_env:default chain_list_prev:prev per_env_again:default chain_list_again:prev two_launch:default:R2S_PF=0 two_launch_prev:prev:R2S_PF=0 > $out/variant_sloth.txt 2>&1; tail -7 $out/variant_sloth.txt
