#!/bin/bash
# counter passes on the final sources, then (r5_final2.sh) the headline with them
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profiling/pmc_r5.sh > /dev/null 2>&1
cp gpurun_out/pmc_r5/r5_pmc_summary.json profiles/r5_pmc_summary.json
bash tools/profiling/r5_final2.sh
