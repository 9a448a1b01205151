#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s10; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 900 python tools/profiling/variant_bench.py head16_noself:default head32:default:R2S_PF_HEAD=32 head16_noself_again:default head32_again:default:R2S_PF_HEAD=32 head8:default:R2S_PF_HEAD=8 > $out/variant_sloth.txt 2>&1; tail -6 $out/variant_sloth.txt
