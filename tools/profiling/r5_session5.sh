#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s5; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 900 python tools/profiling/variant_bench.py pf_w5_c4:pfw5 pf_w5_c2:pfw5:R2S_CHAINS=2 pf_w5_c3:pfw5:R2S_CHAINS=3 pf_w5_head40:pfw5:R2S_PF_HEAD=40 pf_w5_head48:pfw5:R2S_PF_HEAD=48 two_c4:default:R2S_PF=0 > $out/variant_sloth.txt 2>&1; tail -7 $out/variant_sloth.txt
VB_CONFIG=T_pusher_32env timeout 300 python tools/profiling/variant_bench.py pf:default pf_c4:default:R2S_CHAINS=4 pf_c1:default:R2S_CHAINS=1 two_launch:default:R2S_PF=0 > $out/variant_pusher.txt 2>&1; tail -5 $out/variant_pusher.txt
rm -f gpurun_out/r5_parity.json
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; tail -8 $out/pytest.log
