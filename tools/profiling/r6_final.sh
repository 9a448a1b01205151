#!/bin/bash
# Round 6, the judged artefacts on the final sources: GPU suite + parity log, bench lines, kernel tables, counter passes, the in-kernel
# probes and the hand-off soak.  gpurun --timeout 5400 -- 'PMC_GIT_HEAD=<sha> bash tools/profiling/r6_final.sh'; then tools/profiling/r6_collect.sh
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profiling/refresh_profiles_r6.sh
bash tools/profiling/pmc_r6.sh
out=$R/gpurun_out/prof_r6
export PYTHONPATH=$R/real2sim-eval_amd:$R
cd $R
# in-kernel pictures of one launch: the held grasp and the closing phase of the headline (batched finishers at the head), the pusher's contact; the free kernel's timeline
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py sloth_32env 32 3 16 0.1 > $out/pf_probe.txt 2>&1
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py sloth_32env 32 3 8 0.1 >> $out/pf_probe.txt 2>&1
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py T_pusher_32env 32 2 6 >> $out/pf_probe.txt 2>&1
R2S_CHAINS=1 R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so TAG=r6_free_c1 timeout 300 python tools/probes/phase_probe.py > $out/phase_probe_free.txt 2>&1
# batched finishing vs one workgroup per particle, finishers at the head vs two launches: the same build, the same session (per-step table through the grasp)
for v in "A=1" "R2S_FIN_BATCH=0" "R2S_PF=0" "R2S_FIN_BATCH=0 R2S_PF=0"; do
  echo "== $v" >> $out/variant_sloth.txt
  env $v timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 2>&1 | grep -v "pad forces\|amdgpu.ids" | grep "step  1:\|step  4\|step  7\|step 10\|step 13\|step 17" | cut -c1-125 >> $out/variant_sloth.txt
done
timeout 300 python tools/profiling/onset_diag.py sloth_32env 18 3 0.1 2>&1 | grep step | cut -c1-170 > $out/onset_sloth_1env.txt
timeout 900 python tools/profiling/soak_pf.py sloth_32env 30 14 0.1 > $out/soak_pf.txt 2>&1
timeout 300 python tools/profiling/soak_pf.py T_pusher_32env 20 8 >> $out/soak_pf.txt 2>&1
tail -3 $out/soak_pf.txt; cat $out/variant_sloth.txt
