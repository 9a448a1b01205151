#!/bin/bash
# after r5_final.sh + r5_collect.sh + commit: the headline line once more, now with the committed round-5 counter summary (not stale), and the probes
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r5; mkdir -p $out
cd $R; export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_sloth_32env.json 2> $out/bench.err; tail -c 300 $out/bench.err
timeout 300 python bench.py --config rope_1env --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_rope.json
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py sloth_32env 32 2 6 2>&1 | grep -v amdgpu.ids > $out/pf_probe.txt
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/pf_probe.py T_pusher_32env 32 2 6 2>&1 | grep -v amdgpu.ids >> $out/pf_probe.txt
cat $out/pf_probe.txt
python -c "
import json
d=json.loads(open('$out/bench_sloth_32env.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value',d['value'],'stale',r['counters_stale'],'traffic',r['traffic'],'frac',r['frac'],'sync',d['synchronised_window']['env_steps_per_s'],d['synchronised_window']['enqueue_only_env_steps_per_s'])
e=json.loads(open('$out/bench_rope.json').read()); print('rope',e['value'])"
