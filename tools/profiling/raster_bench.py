import sys, os, json, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "real2sim-eval_amd"))
import torch
from r2s_hip.rollout import BatchedRollout
cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
n_env = int(os.environ["NENV"]) if os.environ.get("NENV") else None   # frames per batch = 2 x n_env (side + wrist camera)
ro = BatchedRollout(cfg, num_substeps=2, self_collision=False, n_env=n_env)
ro.step()
ro.raster.set_timing(True)
acc = {}
for i in range(6):
    ro.render(); torch.cuda.synchronize()
    st = ro.raster.stage_ms()
    if i: 
        for k, v in st.items(): acc[k] = acc.get(k, 0) + v / 5
d = ro.raster.debug(); slots = int(d["point_offsets"][-1]) if len(d["point_offsets"]) else 0
h = hashlib.sha1(ro.out_color.cpu().numpy().tobytes() + ro.out_depth.cpu().numpy().tobytes()).hexdigest()[:12]
print(json.dumps({"cfg": cfg, "n_env": ro.n_env, "L": int(ro.last_num_rendered), "slots": slots, "sha": h, "total": sum(acc.values()), **{k: round(v, 4) for k, v in acc.items()}}))
