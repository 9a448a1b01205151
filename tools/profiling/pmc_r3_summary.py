"""Per-kernel means of the rocprofv3 --pmc passes of tools/profiling/pmc_r3.sh -> r3_pmc_summary.json (the file bench.py labels its
`traffic` / `valu_busy_frac` fields with).  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, checked on the
512 MiB copy of the same run), both are in KiB; SQ_* counters count quad-cycles summed over the chip's 1024 SIMDs."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {"k_substep_pf<256, 1024, false, 1, 4>": "k_substep_pf", "k_substep_pf<256, 1024, true, 1, 4>": "k_substep_pf_contact",   # round 6: the batched small-scene finishers at the head
           "k_contact_finish_batch<true>": "k_contact_finish", "k_contact_finish_batch<false>": "k_contact_finish_mesh_only",
           "k_substep_pf<256, 1024, false, 1, 3>": "k_substep_pf", "k_substep_pf<256, 1024, true, 1, 3>": "k_substep_pf_contact",
           "k_substep_pf<256, 1024, false, 2, 2>": "k_substep_pf_large_mesh",
           "k_substep<256, 1024, false, 1>": "k_substep", "k_substep<256, 1024, true, 1>": "k_substep_contact", "k_contact_finish<3, true>": "k_contact_finish",
           "k_contact_finish<3, false>": "k_contact_finish_mesh_only", "k_composite": "k_composite", "k_steps_resident": "k_steps_resident", "k_emit_keys": "k_emit_keys", "k_preprocess": "k_preprocess",
           "k_skin": "k_skin", "k_bone_fit": "k_bone_fit", "k_candidates_fine": "k_candidates_fine", "k_tile_ranges": "k_tile_ranges",
           "k_bin_scatter": "k_bin_scatter", "k_bin_hist": "k_bin_hist",   # round 6: one-pass tile binning
           "direct_copy_kernel": "calibration_copy_512MiB", "__amd_rocclr_copyBuffer": "calibration_copy_512MiB"}


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            for pat, key in KERNELS.items():
                if pat in name:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
    out = {}
    for key, cs in acc.items():
        if key.startswith("calibration"):   # torch launches many small copies too: keep the 512 MiB ones (the three largest per counter)
            cs = {c: sorted(v)[-3:] for c, v in cs.items()}
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        ent = {"dispatches": len(next(iter(cs.values()))), "counters_mean_per_dispatch": {c: round(v, 1) for c, v in sorted(m.items())}}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            ent["fetch_size_kib"], ent["write_size_kib"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
            ent["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        if "SQ_ACTIVE_INST_VALU" in m and m.get("SQ_BUSY_CYCLES", 0) > 0:
            span = m["SQ_BUSY_CYCLES"] / 32.0                      # summed over the 32 shader engines -> cycles the kernel was on the chip
            ent["kernel_span_cycles"] = round(span, 1)
            # a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles (quarter-rate ops longer: counted once, so this
            # is a floor): fraction of the chip's 1024 x span SIMD-cycles spent issuing VALU instructions, <= 1 by construction
            if "SQ_INSTS_VALU" in m:
                ent["valu_busy_frac"] = round(min(1.0, m["SQ_INSTS_VALU"] * 4 / 1024 / span), 4)
            # round 2 reported this ratio under the name valu_busy_frac: VALU-active cycles summed over WAVES; instructions of
            # different waves overlap in the pipeline, so it exceeds 1 on a saturated kernel (not a fraction)
            ent["valu_active_wave_cycles_per_simd_cycle"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / span, 4)
            if "SQ_LDS_IDX_ACTIVE" in m:
                ent["lds_busy_frac"] = round(m["SQ_LDS_IDX_ACTIVE"] / 256 / span, 4)          # LDS-array cycles over 256 CUs
                ent["lds_bank_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m["SQ_LDS_IDX_ACTIVE"], 1), 4)
            if "SQ_WAVE_CYCLES" in m:
                ent["wave_cycles_waiting_frac"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 4)
                ent["wave_cycles_issue_stalled_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4)
        if "TCC_EA0_RDREQ_DRAM_sum" in m or "TCC_EA0_RDREQ_sum" in m:   # the L2's requests to the fabric, and the part of them that went on to DRAM (the rest hit the Infinity Cache)
            ent["l2_fabric_requests"] = {c: m[c] for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_DRAM_sum") if c in m}
            if m.get("TCC_EA0_RDREQ_sum", 0) > 0 and "TCC_EA0_RDREQ_DRAM_sum" in m:
                ent["read_requests_reaching_dram_frac"] = round(m["TCC_EA0_RDREQ_DRAM_sum"] / m["TCC_EA0_RDREQ_sum"], 4)
        if m.get("TCC_REQ_sum", 0) > 0 and "TCC_HIT_sum" in m:
            ent["l2_hit_frac"] = round(m["TCC_HIT_sum"] / m["TCC_REQ_sum"], 4)
        out[key] = ent
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "real2sim-eval_amd"))
    from r2s_hip._lib import kernel_source_sha16
    res = {"source_sha16": kernel_source_sha16(), "git_head": os.environ.get("PMC_GIT_HEAD", "unknown (no .git on the GPU box; pass PMC_GIT_HEAD)"),
           "note": "rocprofv3 --pmc passes (tools/profiling/pmc_r3.sh / pmc_r4.sh) over tools/profiling/pmc_run.py on one MI355X: " + os.environ.get("PMC_CONFIG", "sloth_32env") + ", R2S_CHAINS=1 (a k_substep dispatch = "
                   "one batched substep of all 32 envs), " + (os.environ.get("PMC_STEPS", "5") + " env steps through the closing ramp into the held grasp (PMC_CLOSE_RATE)" if os.environ.get("PMC_CLOSE_RATE") else "2 free + 3 contact env steps") + ".  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                   "(FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md; the 512 MiB calibration copy of the same run is listed).  "
                   "valu_busy_frac = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 shader engines): the share of SIMD issue cycles taken by VALU instructions (a floor: quarter-rate instructions count 4 cycles too), at most 1; lds_busy_frac = SQ_LDS_IDX_ACTIVE / 256 CUs "
                   "over the same span.",
           os.environ.get("PMC_CONFIG", "sloth_32env"): out}
    json.dump(res, open(os.path.join(root, os.environ.get("PMC_SUMMARY_NAME", "r3_pmc_summary.json")), "w"), indent=1)
    for k, e in out.items():
        print(k, {a: b for a, b in e.items() if a != "counters_mean_per_dispatch"})


if __name__ == "__main__":
    main()
