#!/bin/bash
# round 6, session 2: where the suite stands (no -x), deferred counts through the grasp at 32 envs, in-kernel picture of a launch in the hold state
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
R2S_DIAG_DEFER=1 R2S_DIAG_CAND=1 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 > gpurun_out/r6_s2_grasp_diag_sloth32.log 2>&1
R2S_HIP_LIB=scratch/variants/libr2s_probe.so timeout 300 python tools/probes/pf_probe.py sloth_32env 32 3 16 0.1 > gpurun_out/r6_s2_pf_probe_hold.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6_s2_gputests.log 2>&1
echo "gpu tests rc $?" >> gpurun_out/r6_s2_gputests.log
tail -15 gpurun_out/r6_s2_gputests.log
