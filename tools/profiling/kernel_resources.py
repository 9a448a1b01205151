"""Register / LDS / occupancy report of the physics kernels (hipcc -Rpass-analysis=kernel-resource-usage); runs without a GPU.
usage: python tools/profiling/kernel_resources.py [extra hipcc flags, e.g. -DR2S_PIPE]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(R, "real2sim-eval_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "physics.hip")
flags = [a for a in sys.argv[1:] if not a.endswith(".hip")]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", *flags,
                      "-c", src, "-o", "/tmp/_kr.o"], capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}
    for k, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                   ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, l)
        if m and cur:
            rows[cur][k] = int(m.group(1))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    if any(s in name for s in ("k_substep", "contact_finish", "self_finish", "k_composite", "k_steps_resident", "k_emit", "k_preprocess")):
        print(f"{name:60s} {v}")
