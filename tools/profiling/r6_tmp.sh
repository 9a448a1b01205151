cd /root/repo
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
timeout 900 python -m pytest tests/test_fin_batch_gpu.py tests/test_contact_flavours_gpu.py tests/test_flavour_pairs_gpu.py tests/test_pf_gpu.py -m gpu -q -x 2>&1 | tail -3
run() { echo "== $*"; env "$@" timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 2>&1 | grep -v "pad forces" | grep "step  2\|step  7\|step 17" | cut -c1-110; }
run A=1
run A=2
timeout 400 python bench.py --config T_pusher_32env --steps 20 --warmup 5 --no-cpu-baseline --episodes 0 2>/dev/null | tail -1 > gpurun_out/r6_tmp_T.json
python - <<P
import json
d=json.loads(open('gpurun_out/r6_tmp_T.json').read())
print('T_pusher', round(d['value'],1), 'sync', round(d['synchronised_window']['env_steps_per_s'],1), {k:(round(v['substep_us'],2), v['mesh_contacts']) for k,v in d['phases'].items() if isinstance(v,dict)})
P
