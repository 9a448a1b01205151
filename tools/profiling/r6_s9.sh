# session 9: raster front end after the one-pass binning (tests, stage times with both paths, kernel table)
cd /root/repo
timeout 600 python -m pytest tests/test_raster_gpu.py tests/test_parity_round2_gpu.py tests/test_wrist_camera_gpu.py tests/test_scene_files_gpu.py -m gpu -q -x 2>&1 | tail -3
for cfg in sloth_32env rope_1env; do timeout 300 python tools/profiling/raster_bench.py $cfg 2>&1 | tail -1 | cut -c1-260; done
R2S_RASTER_RADIX_SORT=1 timeout 300 python tools/profiling/raster_bench.py sloth_32env 2>&1 | tail -1 | cut -c1-260
bash tools/profiling/r6_raster_trace.sh 2>&1 | grep "k_bin\|k_tile\|k_emit\|k_prepro\|rocprim"
