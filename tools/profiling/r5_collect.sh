#!/bin/bash
# copy what should be judged from gpurun_out/ (scratch) to profiles/ (tracked), named per round
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); s=$R/gpurun_out/prof_r5; d=$R/profiles
cp $s/bench_sloth_32env.json $d/r5_bench_sloth_32env.json
cp $s/bench_other_configs.jsonl $d/r5_bench_other_configs.jsonl
cp $s/kernel_stats_default.md $d/r5_bench_kernel_stats_4chains.md
cp $s/kernel_stats_chains1.md $d/r5_bench_kernel_stats_chains1.md
cp $s/kernel_stats_pusher.md $d/r5_bench_kernel_stats_pusher.md
cp $s/kernel_stats_rope_1env.md $d/r5_bench_kernel_stats_rope_1env.md
cp $s/r5_parity.json $d/r5_parity.json
cp $R/gpurun_out/pmc_r5/r5_pmc_summary.json $d/r5_pmc_summary.json
cp $s/pf_probe.txt $d/r5_pf_probe.txt
cp $s/query_probe_pusher.txt $d/r5_query_probe_pusher.txt
cat $s/variant_sloth.txt $s/variant_pusher.txt | grep -v amdgpu.ids > $d/r5_variant_bench.txt
grep -v amdgpu.ids $s/soak_pf.txt > $d/r5_soak_pf.txt
tail -2 $s/pytest.log
ls -la $d/r5_*
