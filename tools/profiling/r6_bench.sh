#!/bin/bash
# round 6: the driver's bench line + kernel trace of the same command
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_sloth_32env.json 2> gpurun_out/r6_bench_sloth_32env.err
echo "bench rc $?" >> gpurun_out/r6_bench_sloth_32env.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r6_bench_sloth_32env.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print('window substep us', d['window']['substep_us_per_step'])
print('grasped', d['window']['grasped_envs_per_step'])
print('phases', {k:(v['substep_us'], v['mesh_contacts'], v['grasped_envs']) for k,v in d['phases'].items() if isinstance(v,dict)})
print('sync', d['synchronised_window']['env_steps_per_s'], d['synchronised_window']['enqueue_only_env_steps_per_s'])
print('raster', d['raster']['stage_ms'], d['raster']['gs_raster_mpix_per_s'])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us') if k in d['roofline']})
print('episodes', json.dumps(d.get('episodes'))[:1500])
print('gate', d['parity_gate'].get('passed'), d['parity_gate'].get('x_max_abs'))
P
