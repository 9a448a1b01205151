"""Physics-kernel experiment driver (GPU box): for each (label, lib, env) runs a child process that builds the headline scene,
checks the smoke parity of the physics path against the oracle on a small scene with the SAME layout knobs, and times the
captured 667-substep graph in free flight (far from the gripper: no finishing kernel; and with the idle finishing kernel forced
into the graph) and in the grasp.  One line per variant.
usage: python tools/profiling/variant_bench.py label:lib[:K=V,...] ..."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json
R = %r
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R, os.path.join(R, "tests")]
import numpy as np, torch
from r2s_hip.rollout import BatchedRollout
from util_physics import hip_env, make_object, oracle_env
out = {}
# parity smoke with the same knobs: 3000-particle sloth (several blocks + halo) with ground contact, 60 substeps
ob = make_object("sloth", 3000, seed=1, lift=0.0005)
ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = -0.3
o = oracle_env(ob, num_substeps=60, self_collision=False); h = hip_env(ob, num_substeps=60, self_collision=False, n_env=3)
o.step(); h.step()
out["parity_max_abs"] = float(np.abs(h.x.cpu().numpy() - o.x[None]).max())
out["layout_small"] = h.layout_stats()
del h
cfg = os.environ.get("VB_CONFIG", "sloth_32env")
ro = BatchedRollout(cfg, num_substeps=667, close_at=6)
out["layout"] = ro.phys.layout_stats()
ro.phys.set_timing(True)
def t(n=3):
    v = []
    for _ in range(n):
        ro.physics_step(); ro.t += 1
        torch.cuda.synchronize()
        ms, k = ro.phys.last_step_ms(); v.append(ms / k * 1e3)
    return min(v), ro.phys.last_flavour()["kernel"]
out["free_far_us"], out["free_far_flavour"] = t(3)          # gripper > 3 cm above the arms: one kernel per substep
ro.phys.set_tuning(chains=int(os.environ.get("R2S_CHAINS", "0")), mesh_defer=1)
out["free_idle_finish_us"], _ = t(3)                         # the same with the idle finishing launch in the graph
ro.phys.set_tuning(chains=int(os.environ.get("R2S_CHAINS", "0")), mesh_defer=-1)
while ro.t < 8:
    ro.physics_step(); ro.t += 1
out["contact_us"], out["contact_flavour"] = t(3)             # in the grasp
st = ro.contact_stats()
out["contact_candidates"], out["contact_mesh_hits"] = st["self_collision_candidates"], st["mesh_contacts"]
print("VB_RESULT " + json.dumps(out))
''' % R
rows = []
for spec in sys.argv[1:]:
    parts = spec.split(":")
    label, lib = parts[0], parts[1]
    env = dict(os.environ)
    if lib not in ("", "default"):
        env["R2S_HIP_LIB"] = os.path.join(R, "scratch", "variants", f"libr2s_{lib}.so")
    if len(parts) > 2 and parts[2]:
        for kv in parts[2].split(","):
            k, v = kv.split("=")
            env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("VB_RESULT ")]
    if not line:
        print(label, "FAILED", r.stderr[-600:])
        continue
    d = json.loads(line[0][10:])
    rows.append((label, d))
    lay = d["layout"]
    print(f"{label:28s} parity {d['parity_max_abs']:.1e}  blocks {lay['blocks']:3d} halo {lay['halo_max']:4d} fallback {lay['fallback_slots']:6d} chains {lay['chains']}  "
          f"free {d['free_far_us']:6.2f}  free+idle-finish {d['free_idle_finish_us']:6.2f}  contact {d['contact_us']:6.2f} us/substep "
          f"({d['contact_candidates']} cand, {d['contact_mesh_hits']} hits) {d['contact_flavour']}", flush=True)
json.dump(rows, open(os.path.join(R, "gpurun_out", "variant_bench.json"), "w"), indent=1)
