#!/bin/bash
# bench lines of the round: the headline as the driver runs it + the other configs. gpurun --timeout 1800 -- 'bash tools/profiling/r5_bench.sh'
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_bench; mkdir -p $out
cd $R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_sloth_32env.json 2> $out/bench.err; tail -c 400 $out/bench.err
: > $out/bench_other_configs.jsonl
for cfg in rope_1env T_pusher_32env sloth_multicam_8env rope_fold_1env; do
  timeout 400 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
done
# the folded rope with the per-substep kernels (what a small batch with live candidates ran until round 4)
R2S_RES_SELF=0 timeout 400 python bench.py --config rope_fold_1env --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | tail -1 > $out/bench_rope_fold_per_substep_kernels.json
python - <<'PY'
import json
rows = [json.loads(open('gpurun_out/r5_bench/bench_sloth_32env.json').read().strip().splitlines()[-1])] + [json.loads(l) for l in open('gpurun_out/r5_bench/bench_other_configs.jsonl')]
for d in rows:
    ph = {k: (round(v['ms_per_step'], 2), round(v['substep_us'], 2), v['mesh_contacts'], v['self_collision_candidates']) for k, v in d['phases'].items() if isinstance(v, dict)}
    print(d['config']['workload'][:40], 'value', None if d['value'] is None else round(d['value'], 1), 'sync', round(d.get('synchronised_window', {}).get('env_steps_per_s', 0), 1), ph,
          'closed/enq', round(d['closed_loop_get_obs']['ratio'], 3), 'pipelined', round(d.get('throughput_mode', {}).get('pipelined_env_steps_per_s', 0), 1), 'gate', d['parity_gate'].get('passed'),
          'window us', d['window']['substep_us_per_step'][:3], d['window']['substep_us_per_step'][-3:], 'raster ms', round(sum(d['raster']['stage_ms'].values()), 3))
PY
