cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c12; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c12/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print({k:(v['ms_per_step'],v['physics_ms_per_step'],v['substep_us']) for k,v in d['phases'].items() if isinstance(v,dict)}); print(d['raster']['stage_ms'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/trace.log 2>&1 || echo trace-failed
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/profiling/gaps.py $db > $out/gaps.txt 2>&1; tail -8 $out/gaps.txt | cut -c1-300
rm -rf $out/trace
