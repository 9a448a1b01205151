#!/bin/bash
# kernel-trace stats of the folded rope (the resident stepper's self-collision flavour): one k_steps_resident<512,true,1> launch per env step
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r5; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace_rope_fold -o bench -- python $R/bench.py --config rope_fold_1env --steps 8 --warmup 5 --no-cpu-baseline --no-parity-gate --no-pipelined > $out/bench_trace_rope_fold.log 2>&1 || echo trace-failed
db=$(find $out/trace_rope_fold -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $out/kernel_stats_rope_fold_1env.md > /dev/null 2>&1 || echo stats-failed
head -8 $out/kernel_stats_rope_fold_1env.md | cut -c1-80,150-260
rm -rf $out/trace_rope_fold
