cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c7; mkdir -p $out
R2S_DIAG_CAND=1 timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 4 9 > $out/diag32.log 2>&1; tail -12 $out/diag32.log | cut -c1-150
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c7/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['phases'],indent=0)); print(d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['single_chain_check']); print(d['raster']['stage_ms']); print(d['cpu_baseline'])
PY
