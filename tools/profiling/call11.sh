cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c11; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 300 python bench.py --config sloth_multicam_8env --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_mc.json 2> $out/bench_mc.err; tail -3 $out/bench_mc.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c11/bench_mc.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['raster']['gs_raster_mpix_per_s'], d['raster']['stage_ms'], d['skinning_ms_per_env_step'], d['physics_ms_per_env_step'])
PY
