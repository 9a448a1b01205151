#!/bin/bash
# refresh the judged artifacts: kernel-trace stats of the bench command (default = 2 concurrent chains, and R2S_CHAINS=1
# where per-kernel durations do not overlap) + FETCH/WRITE traffic (separate --pmc passes)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_final; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in default chains1; do
  if [ $mode = chains1 ]; then export R2S_CHAINS=1; else unset R2S_CHAINS; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace_$mode -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_trace_$mode.log 2>&1 || echo trace-failed
  db=$(find $out/trace_$mode -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $db $out/kernel_stats_$mode.md > /dev/null 2>&1 || echo stats-failed
  head -4 $out/kernel_stats_$mode.md | cut -c1-150; tail -1 $out/bench_trace_$mode.log | cut -c1-200
  rm -rf $out/trace_$mode
done
unset R2S_CHAINS
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p1 -o p1 -- python $R/tools/profiling/traffic_run.py > $out/p1.log 2>&1 || echo fail1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/p2 -o p2 -- python $R/tools/profiling/traffic_run.py > $out/p2.log 2>&1 || echo fail2
