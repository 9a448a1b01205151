#!/bin/bash
# Round-5 counter passes over tools/profiling/pmc_run.py: SQ activity (3 passes), FETCH_SIZE and WRITE_SIZE (one pass each, as
# the microarchitecture guide prescribes), then tools/profiling/pmc_r3_summary.py -> gpurun_out/pmc_r5/r5_pmc_summary.json (PMC_SUMMARY_NAME)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_r5; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export R2S_CHAINS=1
export PMC_SUMMARY_NAME=r5_pmc_summary.json
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 280 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o p$i -- python $R/tools/profiling/pmc_run.py > $out/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $out/p$i.log)"
done
cd $R; python tools/profiling/pmc_r3_summary.py $out > $out/summary.log 2>&1; tail -40 $out/summary.log
# keep only the summaries (the raw CSVs are hundreds of MB)
find $out -name "*.csv" -size +2M -delete
