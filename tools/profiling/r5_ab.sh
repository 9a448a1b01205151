#!/bin/bash
# A/B of kernel variants on the headline: product build and scratch/variants/libr2s_<name>.so, alternating; contact / free substep and value
# usage (through gpurun): bash tools/profiling/r5_ab.sh name [name ...]
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for v in product "$@"; do
    if [ $v = product ]; then unset R2S_HIP_LIB; else export R2S_HIP_LIB=$R/scratch/variants/libr2s_$v.so; fi
    timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pipelined --no-parity-gate 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ph=d['phases']
print('$v', 'value', round(d['value'],1), 'sync', round(d['synchronised_window']['env_steps_per_s'],1), 'free/contact us', round(ph['free']['substep_us'],2), round(ph['contact']['substep_us'],2), 'window', [round(x,2) for x in d['window']['substep_us_per_step'][-4:]])"
  done
done
