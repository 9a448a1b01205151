#!/bin/bash
# Round-6 counter passes over tools/profiling/pmc_run.py (the closing ramp: the run ends in the HELD grasp, so k_substep_pf<..., true, 1, 4> is the contact flavour an
# episode spends its time in): SQ activity (3 passes), FETCH_SIZE and WRITE_SIZE (one pass each, as the microarchitecture guide prescribes), a pass of the L2's
# DRAM-side request counters if this rocprofv3 exposes them, then tools/profiling/pmc_r3_summary.py -> gpurun_out/pmc_r6/r6_pmc_summary.json
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_r6; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export R2S_CHAINS=1 PMC_CLOSE_RATE=0.1 PMC_STEPS=14
export PMC_SUMMARY_NAME=r6_pmc_summary.json
rocprofv3 --list-avail 2>/dev/null | grep -i "DRAM\|TCC_EA0\|MALL\|TCC_HIT\|TCC_MISS\|TCC_REQ\b" | head -60 > $out/avail_l2_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o p$i -- python $R/tools/profiling/pmc_run.py > $out/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $out/p$i.log)"
done
cd $R; python tools/profiling/pmc_r3_summary.py $out > $out/summary.log 2>&1; tail -40 $out/summary.log
# keep only the summaries (the raw CSVs are hundreds of MB)
find $out -name "*.csv" -size +2M -delete
