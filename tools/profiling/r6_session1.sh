#!/bin/bash
# round 6, session 1: the whole GPU suite on the new sources (closed finger meshes, flavour function, closing ramp), the driver's bench line, a grasp trace
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_grasp_closed_loop_gpu.py > gpurun_out/r6_s1_gputests.log 2>&1
echo "gpu tests rc $?" >> gpurun_out/r6_s1_gputests.log
timeout 900 python -m pytest tests/test_grasp_closed_loop_gpu.py tests/test_flavour_pairs_gpu.py -m gpu -q > gpurun_out/r6_s1_grasp_tests.log 2>&1
echo "grasp tests rc $?" >> gpurun_out/r6_s1_grasp_tests.log
timeout 300 python tools/profiling/grasp_diag.py sloth_32env 4 3 18 0.1 > gpurun_out/r6_s1_grasp_diag_sloth.log 2>&1
timeout 300 python tools/profiling/grasp_diag.py rope_1env 1 3 18 0.1 > gpurun_out/r6_s1_grasp_diag_rope.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_s1_bench_sloth.json 2> gpurun_out/r6_s1_bench_sloth.err
echo "bench rc $?" >> gpurun_out/r6_s1_bench_sloth.err
tail -3 gpurun_out/r6_s1_gputests.log; tail -3 gpurun_out/r6_s1_grasp_tests.log
