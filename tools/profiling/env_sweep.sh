#!/bin/bash
# Headline scene at 8 … 128 environments per GPU (one MI355X): is 32 per GPU — the reference's figure — where the chip saturates?
# Each line is bench.py's own JSON (parity gate on, no CPU baseline, no throughput-mode comparison).  Output: gpurun_out/env_sweep.jsonl
mkdir -p gpurun_out
: > gpurun_out/env_sweep.jsonl
for E in ${ENVS:-8 16 32 64 128}; do
    timeout ${SWEEP_TIMEOUT:-150} python bench.py --envs $E --steps ${STEPS:-10} --warmup 4 --no-cpu-baseline --no-pipelined 2> gpurun_out/env_sweep_$E.err | tail -1 >> gpurun_out/env_sweep.jsonl
    echo "envs $E: exit ${PIPESTATUS[0]}"
done
python - <<'PY'
import json
print("| envs per GPU | env-steps/s | ms per step | free substep µs | contact substep µs | raster ms | chains |")
print("|---|---|---|---|---|---|---|")
for l in open("gpurun_out/env_sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    if "value" not in d or d["value"] is None: continue
    ph, r = d.get("phases", {}), d.get("raster", {})
    st = r.get("stage_ms", {})
    print(f"| {d['config']['envs_per_gpu']} | {d['value']:.0f} | {d['ms_per_step']:.2f} | {ph.get('free',{}).get('substep_us',0):.1f} | {ph.get('contact',{}).get('substep_us',0):.1f} | {sum(st.values()):.2f} | {d['roofline'].get('concurrent_chains')} |")
PY
