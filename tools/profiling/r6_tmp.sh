cd /root/repo
export R2S_PARITY_LOG=gpurun_out/r6_parity.json
timeout 1200 python -m pytest tests/test_contact_flavours_gpu.py tests/test_flavour_pairs_gpu.py tests/test_grasp_closed_loop_gpu.py tests/test_resident_gpu.py tests/test_pf_gpu.py -m gpu -q 2>&1 | tail -15
