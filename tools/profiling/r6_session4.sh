#!/bin/bash
# round 6, session 4: the whole GPU suite on the batched finisher + lag 1 for small batches; onset trace of the one-environment toy
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
python tools/profiling/onset_diag.py sloth_32env 18 3 0.1 2>&1 | grep step > gpurun_out/r6_s4_onset_sloth_1env.txt
cat gpurun_out/r6_s4_onset_sloth_1env.txt | cut -c1-150
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6_s4_gputests.log 2>&1
echo "gpu tests rc $?" >> gpurun_out/r6_s4_gputests.log
tail -25 gpurun_out/r6_s4_gputests.log
