#!/bin/bash
# round 6, session 3: batched small-scene finishing — bit identity with the old form, timing through the grasp at 32 envs, probe
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
timeout 900 python -m pytest tests/test_fin_batch_gpu.py -m gpu -q -x > gpurun_out/r6_s3_fin_batch_tests.log 2>&1
echo "rc $?" >> gpurun_out/r6_s3_fin_batch_tests.log
tail -30 gpurun_out/r6_s3_fin_batch_tests.log
R2S_DIAG_DEFER=1 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 2>&1 | grep -v "pad forces" > gpurun_out/r6_s3_grasp_diag_sloth32.log
R2S_FIN_BATCH=0 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 2>&1 | grep -v "pad forces" > gpurun_out/r6_s3_grasp_diag_sloth32_old.log
tail -4 gpurun_out/r6_s3_grasp_diag_sloth32.log; tail -2 gpurun_out/r6_s3_grasp_diag_sloth32_old.log
R2S_HIP_LIB=scratch/variants/libr2s_probe.so timeout 300 python tools/probes/pf_probe.py sloth_32env 32 3 16 0.1 > gpurun_out/r6_s3_pf_probe_hold.txt 2>&1
cat gpurun_out/r6_s3_pf_probe_hold.txt
