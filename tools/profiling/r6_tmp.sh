cd /root/repo
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
timeout 900 python -m pytest tests/test_fin_batch_gpu.py tests/test_pf_gpu.py tests/test_contact_flavours_gpu.py tests/test_flavour_pairs_gpu.py -m gpu -q -x 2>&1 | tail -3
run() { echo "== $*"; env "$@" timeout 300 python tools/profiling/grasp_diag.py sloth_32env 32 3 18 0.1 2>&1 | grep -v "pad forces" | grep "step  2:\|step  7\|step 17" | cut -c1-110; }
run R2S_FIN_SLOTS=16
run R2S_FIN_SLOTS=12
run R2S_FIN_SLOTS=20
run R2S_FIN_SLOTS=24
run R2S_HIP_LIB=scratch/variants/libr2s_np.so
