#!/bin/bash
# the GPU test suite with the parity log of the round: gpurun --timeout 2400 -- 'bash tools/profiling/r5_gputests.sh [pytest args]'
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_tests; mkdir -p $out
cd $R
rm -f gpurun_out/r5_parity.json gpurun_out/r4_parity.json
timeout 2000 python -m pytest tests -m gpu -q "$@" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
grep -E "passed|failed|FAILED|ERROR|rc " $out/pytest.log | tail -30
