#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_s11; rm -rf $out; mkdir -p $out
cd $R
export PYTHONPATH=$R/real2sim-eval_amd:$R
timeout 900 python tools/profiling/variant_bench.py three_b64:default b128_b64:win16 three_b64_again:default b128_b64_again:win16 c1_three_b64:default:R2S_CHAINS=1 c1_b128_b64:win16:R2S_CHAINS=1 > $out/variant_sloth.txt 2>&1; tail -7 $out/variant_sloth.txt
# LDS counters of the free kernel on the variant (one pass)
cd /tmp && export TMPDIR=/tmp
export R2S_CHAINS=1
for lib in default win16; do
  if [ $lib = win16 ]; then export R2S_HIP_LIB=$R/scratch/variants/libr2s_win16.so; else unset R2S_HIP_LIB; fi
  timeout 280 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p_$lib -o p -- python $R/tools/profiling/pmc_run.py > $out/p_$lib.log 2>&1 || echo "pass failed"
  python $R/tools/pmc_summary.py $out/p_$lib "k_substep<256, 1024, false, 1>" | head -8
  find $out/p_$lib -name "*.csv" -size +2M -delete
done
