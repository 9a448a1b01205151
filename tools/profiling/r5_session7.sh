#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s7; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 900 python tools/profiling/variant_bench.py contiguous:default stripe_all:default:R2S_STRIPE=1 stripe_pf:default:R2S_STRIPE=2 two_launch:default:R2S_PF=0 two_launch_stripe:default:R2S_PF=0,R2S_STRIPE=1 > $out/variant_sloth.txt 2>&1; tail -6 $out/variant_sloth.txt
VB_CONFIG=T_pusher_32env timeout 300 python tools/profiling/variant_bench.py contiguous:default stripe_all:default:R2S_STRIPE=1 stripe_pf:default:R2S_STRIPE=2 > $out/variant_pusher.txt 2>&1; tail -4 $out/variant_pusher.txt
R2S_PF=0 R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/query_probe.py T_pusher_32env 32 2 6 > $out/query_probe_pusher.txt 2>&1; tail -6 $out/query_probe_pusher.txt
