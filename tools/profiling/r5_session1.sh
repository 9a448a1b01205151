#!/bin/bash
# Round 5, first GPU session: reference numbers of the round-4 sources on this box + where the finishing launch spends its time.
#   gpurun --timeout 900 -- 'bash tools/profiling/r5_session1.sh'
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s1; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
# 1. free / idle-finish / contact microseconds per batched substep, headline scene and the pusher (default library)
timeout 300 python tools/profiling/variant_bench.py base:default > $out/variant_sloth.txt 2>&1; cat $out/variant_sloth.txt | tail -2
VB_CONFIG=T_pusher_32env timeout 300 python tools/profiling/variant_bench.py base:default > $out/variant_pusher.txt 2>&1; cat $out/variant_pusher.txt | tail -2
# 2. in-kernel stamps of the finishing launch (probe build), default chains: mesh particles (part 1) and the busy candidate wavefronts (part 2)
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/query_probe.py sloth_32env 32 2 6 > $out/query_probe_sloth.txt 2>&1; tail -14 $out/query_probe_sloth.txt
R2S_HIP_LIB=$R/scratch/variants/libr2s_probe.so timeout 200 python tools/probes/query_probe.py T_pusher_32env 32 2 6 > $out/query_probe_pusher.txt 2>&1; tail -8 $out/query_probe_pusher.txt
# 3. candidate counts of the particles with live candidates in the grasp
timeout 200 python - > $out/cand_counts.txt 2>&1 <<'PY'
import numpy as np, torch
from r2s_hip.rollout import BatchedRollout
ro = BatchedRollout("sloth_32env", close_at=2)
for _ in range(6): ro.step()
torch.cuda.synchronize()
num, idx = ro.phys.collision_lists()
n = num.cpu().numpy().reshape(ro.n_env, -1)
live = n[n > 0]
print("particles with candidates", live.size, "per env", (n > 0).sum(1)[:8], "count percentiles 10/50/90/99/max", np.percentile(live, [10, 50, 90, 99, 100]))
print("histogram (1-16, 17-32, 33-64, 65-128, 129-256, 257-500):", [int(((live >= lo) & (live <= hi)).sum()) for lo, hi in ((1,16),(17,32),(33,64),(65,128),(129,256),(257,500))])
print("deferred per substep (last 3):", ro.phys.deferred_counts()[-4:-1], "tagged", ro.phys.tagged_count())
PY
cat $out/cand_counts.txt | tail -4
# 4. raster stages: Gaussian depth sort at 8 bits per pass (5 passes) vs 10 (4 passes)
timeout 200 python tools/profiling/raster_bench.py sloth_32env > $out/raster_base.json 2>/dev/null; cat $out/raster_base.json
R2S_HIP_LIB=$R/scratch/variants/libr2s_gsr10.so timeout 200 python tools/profiling/raster_bench.py sloth_32env > $out/raster_gsr10.json 2>/dev/null; cat $out/raster_gsr10.json
