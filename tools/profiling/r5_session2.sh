#!/bin/bash
# Round 5, session 2: the finishers at the head of the next launch (k_substep_pf) — bit identity against the two-launch form, then timing.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s2; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 600 python -m pytest tests/test_pf_gpu.py -x -q > $out/pytest_pf.log 2>&1; tail -15 $out/pytest_pf.log
timeout 300 python tools/profiling/variant_bench.py pf:default two_launch:default:R2S_PF=0 > $out/variant_sloth.txt 2>&1; tail -3 $out/variant_sloth.txt
VB_CONFIG=T_pusher_32env timeout 300 python tools/profiling/variant_bench.py pf:default two_launch:default:R2S_PF=0 > $out/variant_pusher.txt 2>&1; tail -3 $out/variant_pusher.txt
