cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c14; mkdir -p $out
timeout 200 python tools/profiling/raster_bench.py sloth_32env 2>&1 | tail -1
R2S_NO_TILE_ORDER=1 timeout 200 python tools/profiling/raster_bench.py sloth_32env 2>&1 | tail -1
timeout 200 python tools/profiling/raster_bench.py sloth_multicam_8env 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -k "raster or full_size or randomized or parity" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
