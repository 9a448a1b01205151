#!/bin/bash
# round 6, session 6: counters on the final kernels (pmc_r6.sh), the free kernel's in-launch timeline (one chain), kernel trace of the bench command
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R2S_CHAINS=1 R2S_HIP_LIB=scratch/variants/libr2s_probe.so TAG=r6_free_c1 timeout 300 python tools/probes/phase_probe.py > gpurun_out/r6_phase_probe_free_chains1.txt 2>&1
R2S_HIP_LIB=scratch/variants/libr2s_probe.so TAG=r6_free_c4 NB=472 timeout 300 python tools/probes/phase_probe.py > gpurun_out/r6_phase_probe_free_chains4.txt 2>&1
cat gpurun_out/r6_phase_probe_free_chains1.txt
bash tools/profiling/pmc_r6.sh 2>&1 | tail -30
cat gpurun_out/pmc_r6/avail_l2_counters.txt | head -30
