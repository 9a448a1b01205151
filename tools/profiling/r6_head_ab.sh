# closed-loop cost of launching the chains: one graph per chain (R2S_GRAPH_HEAD=0) vs head + tail graphs, same box
cd /root/repo
for hd in 0 64 0 64; do
  R2S_GRAPH_HEAD=$hd timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --episodes 0 --no-parity-gate --no-pipelined 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d=json.load(open("/tmp/b.json")); sw=d["synchronised_window"]
print("head $hd: value %.1f  ms/step %.3f  sync closed %.1f  enqueue-only %.1f  p50 %.2f max %.2f" % (d["value"], d["ms_per_step"], sw["env_steps_per_s"], sw["enqueue_only_env_steps_per_s"], d["window"]["step_latency_ms"]["p50"], d["window"]["step_latency_ms"]["max"]))
PY
done

