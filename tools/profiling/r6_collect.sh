#!/bin/bash
# copy what should be judged from gpurun_out/ (scratch) to profiles/ (tracked), named per round
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); s=$R/gpurun_out/prof_r6; d=$R/profiles
cp $s/bench_sloth_32env.json $d/r6_bench_sloth_32env.json
cp $s/bench_other_configs.jsonl $d/r6_bench_other_configs.jsonl
cp $s/kernel_stats_default.md $d/r6_bench_kernel_stats_4chains.md
cp $s/kernel_stats_chains1.md $d/r6_bench_kernel_stats_chains1.md
cp $s/kernel_stats_pusher.md $d/r6_bench_kernel_stats_pusher.md
cp $s/kernel_stats_rope_1env.md $d/r6_bench_kernel_stats_rope_1env.md
cp $s/r6_parity.json $d/r6_parity.json
cp $R/gpurun_out/pmc_r6/r6_pmc_summary.json $d/r6_pmc_summary.json
grep -v amdgpu.ids $s/pf_probe.txt > $d/r6_pf_probe.txt
grep -v amdgpu.ids $s/phase_probe_free.txt > $d/r6_phase_probe_free.txt
cp $s/variant_sloth.txt $d/r6_variant_bench.txt
cp $s/onset_sloth_1env.txt $d/r6_onset_sloth_1env.txt
grep -v amdgpu.ids $s/soak_pf.txt > $d/r6_soak_pf.txt
cp $s/raster_kernels.md $d/r6_raster_kernels.md
cp $s/raster_stage_ab.txt $d/r6_raster_stage_ab.txt
cp $s/graph_head_ab.txt $d/r6_graph_head_ab.txt
tail -2 $s/pytest.log
ls -la $d/r6_*
