#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s4; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 600 python -m pytest tests/test_pf_gpu.py -x -q > $out/pytest_pf.log 2>&1; tail -6 $out/pytest_pf.log
timeout 600 python tools/profiling/variant_bench.py pf_w6:default pf_w5:pfw5 pf_w4:pfw4 pf_w6_head16:default:R2S_PF_HEAD=16 pf_w6_head24:default:R2S_PF_HEAD=24 pf_w5_head24:pfw5:R2S_PF_HEAD=24 two_launch:default:R2S_PF=0 > $out/variant_sloth.txt 2>&1; tail -8 $out/variant_sloth.txt
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so R2S_CHAINS=4 timeout 200 python tools/probes/pf_probe.py sloth_32env 32 2 6 > $out/pf_probe_sloth_c4.txt 2>&1; tail -10 $out/pf_probe_sloth_c4.txt
