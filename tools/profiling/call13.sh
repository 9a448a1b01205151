cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c13; mkdir -p $out
timeout 900 python -m pytest tests/test_sink_gpu.py tests/test_abi.py -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -8 $out/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sink /tmp/sink_out > $out/bench_sink.json 2> $out/bench_sink.err; tail -3 $out/bench_sink.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c13/bench_sink.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['observation_sink'])
PY
du -sh /tmp/sink_out; ls /tmp/sink_out/rank0/run/episode_0000/camera_0/rgb | head -3
