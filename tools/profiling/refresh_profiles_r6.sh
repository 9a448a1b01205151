#!/bin/bash
# Refresh the judged artefacts of a round (run on the GPU box through gpurun from the repository root):
#   bench lines (headline as the driver runs it, the other workloads, the observation sink), kernel-trace stats of the bench
#   command with the default chains (four streams) and with one chain (per-kernel durations do not overlap), the GPU test suite with the
#   parity log.  Counter passes: tools/profiling/pmc_r6.sh (separate call).  Copy what should be judged from gpurun_out/ to profiles/.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r6; rm -rf $out; mkdir -p $out
cd $R
rm -f gpurun_out/r6_parity.json
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
cp gpurun_out/r6_parity.json $out/r6_parity.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_sloth_32env.json 2> $out/bench.err; tail -c 600 $out/bench_sloth_32env.json; echo
: > $out/bench_other_configs.jsonl
for cfg in rope_1env T_pusher_32env sloth_multicam_8env rope_fold_1env; do
  timeout 400 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --episodes 0 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
done
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episodes 0 --res 848x480 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl   # the reference's default frame
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episodes 0 --sink /tmp/r6_sink 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/prof_r6/bench_other_configs.jsonl'):
    d = json.loads(l); print(d['config']['workload'][:60], round(d['value'], 1), {k: (round(v['ms_per_step'], 2), v['mesh_contacts'], v['self_collision_candidates']) for k, v in d['phases'].items() if isinstance(v, dict)}, d.get('observation_sink'))
PY
cd /tmp && export TMPDIR=/tmp
for mode in default chains1; do
  if [ $mode = chains1 ]; then export R2S_CHAINS=1; else unset R2S_CHAINS; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace_$mode -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate --no-pipelined --episodes 0 > $out/bench_trace_$mode.log 2>&1 || echo trace-failed
  db=$(find $out/trace_$mode -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $db $out/kernel_stats_$mode.md > /dev/null 2>&1 || echo stats-failed
  head -9 $out/kernel_stats_$mode.md | cut -c1-70,150-240
  rm -rf $out/trace_$mode
done
unset R2S_CHAINS
# configs[1]: one environment — the env step as ONE resident launch, free and (round 5: query servers in the launch) with the gripper on the rope
unset R2S_CHAINS
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace_rope -o bench -- python $R/bench.py --config rope_1env --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate --no-pipelined --episodes 0 > $out/bench_trace_rope.log 2>&1 || echo trace-failed
db=$(find $out/trace_rope -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $out/kernel_stats_rope_1env.md > /dev/null 2>&1 || echo stats-failed
head -5 $out/kernel_stats_rope_1env.md | cut -c1-70,150-240
rm -rf $out/trace_rope
# the large-mesh finishing kernel (k_contact_finish<2>) in its own table: the pusher workload, second half of the window in contact
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace_pusher -o bench -- python $R/bench.py --config T_pusher_32env --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate --no-pipelined --episodes 0 > $out/bench_trace_pusher.log 2>&1 || echo trace-failed
db=$(find $out/trace_pusher -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $out/kernel_stats_pusher.md > /dev/null 2>&1 || echo stats-failed
head -6 $out/kernel_stats_pusher.md | cut -c1-70,150-240
rm -rf $out/trace_pusher
# raster path alone: per-kernel table of the 64-frame batch, stage times with the one-pass binning and with the radix sort (A/B knob), images hashed
cd $R
bash tools/profiling/r6_raster_trace.sh > $out/raster_trace.log 2>&1; cp gpurun_out/r6_raster_kernels.md $out/raster_kernels.md
{ echo "== one-pass binning (build)"; timeout 300 python tools/profiling/raster_bench.py sloth_32env 2>&1 | tail -1;
  echo "== radix sort of the instances (R2S_RASTER_RADIX_SORT=1)"; R2S_RASTER_RADIX_SORT=1 timeout 300 python tools/profiling/raster_bench.py sloth_32env 2>&1 | tail -1;
  echo "== one environment (rope_1env), both"; timeout 300 python tools/profiling/raster_bench.py rope_1env 2>&1 | tail -1; R2S_RASTER_RADIX_SORT=1 timeout 300 python tools/profiling/raster_bench.py rope_1env 2>&1 | tail -1; } > $out/raster_stage_ab.txt
cat $out/raster_stage_ab.txt | cut -c1-250
# closed-loop cost of launching the chains: one graph per chain vs head + tail
bash tools/profiling/r6_head_ab.sh 2>&1 | grep "^head" > $out/graph_head_ab.txt; cat $out/graph_head_ab.txt
