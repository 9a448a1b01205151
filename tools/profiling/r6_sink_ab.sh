# observation sink: D2H on its own stream — sink test, the 20-step window with the sink on, one sustained episode per slot
cd /root/repo
timeout 600 python -m pytest tests/test_sink_gpu.py tests/test_evaluate_gpu.py -m gpu -q -x 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episodes 0 --no-parity-gate --no-pipelined --sink /tmp/r6_sink 2>/dev/null | tail -1 > /tmp/b1.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episodes 1 --no-parity-gate --no-pipelined 2>/dev/null | tail -1 > /tmp/b2.json
python - <<PY
import json
d=json.load(open("/tmp/b1.json")); print("window with sink: value %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("observation_sink"))
d=json.load(open("/tmp/b2.json")); e=d["episodes"]; r=e["runs"][0]
print("value %.1f; episodes sustained %.1f = %.3f x value; latency" % (d["value"], r["sustained_env_steps_per_s"], e["sustained_over_value"]), {k:round(v,2) for k,v in r["step_latency_ms"].items() if k!="steps"}, r["sink"])
PY
