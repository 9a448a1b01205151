# session 8: one-pass tile binning vs the radix sort (tests, stage times, image hash); physics A/B of the per-environment part 2
cd /root/repo
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_parity_round2_gpu.py tests/test_wrist_camera_gpu.py -m gpu -q -x 2>&1 | tail -5
for cfg in sloth_32env rope_1env; do
  echo "== $cfg bin pass"; timeout 300 python tools/profiling/raster_bench.py $cfg 2>&1 | tail -1
  echo "== $cfg radix sort"; R2S_RASTER_RADIX_SORT=1 timeout 300 python tools/profiling/raster_bench.py $cfg 2>&1 | tail -1
done
run() { echo "== $*"; env "$@" timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 2>&1 | grep -v "pad forces" | grep "step  2:\|step  7\|step 17" | cut -c1-110; }
run A=1
run R2S_HIP_LIB=scratch/variants/libr2s_head.so
run A=2
run R2S_HIP_LIB=scratch/variants/libr2s_head.so
