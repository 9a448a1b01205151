#!/bin/bash
# Round 5, the judged artefacts on the final sources: GPU suite + parity log, bench lines, kernel tables, counter passes, the in-kernel
# probes and the hand-off soak.  gpurun --timeout 3400 -- 'PMC_GIT_HEAD=<sha> bash tools/profiling/r5_final.sh'; then tools/profiling/r5_collect.sh
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profiling/refresh_profiles_r5.sh
bash tools/profiling/pmc_r5.sh
out=$R/gpurun_out/prof_r5
export PYTHONPATH=$R/real2sim-eval_amd:$R
cd $R
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py sloth_32env 32 2 6 > $out/pf_probe.txt 2>&1
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py T_pusher_32env 32 2 6 >> $out/pf_probe.txt 2>&1
R2S_PF=0 R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/query_probe.py T_pusher_32env 32 2 6 > $out/query_probe_pusher.txt 2>&1
timeout 300 python tools/profiling/variant_bench.py finishers_at_head:default two_launches:default:R2S_PF=0 > $out/variant_sloth.txt 2>&1
VB_CONFIG=T_pusher_32env timeout 300 python tools/profiling/variant_bench.py finishers_at_head:default two_launches:default:R2S_PF=0 > $out/variant_pusher.txt 2>&1
timeout 600 python tools/profiling/soak_pf.py sloth_32env 30 10 > $out/soak_pf.txt 2>&1
timeout 300 python tools/profiling/soak_pf.py T_pusher_32env 20 8 >> $out/soak_pf.txt 2>&1
tail -3 $out/soak_pf.txt; tail -2 $out/variant_sloth.txt; tail -2 $out/variant_pusher.txt
