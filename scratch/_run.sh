python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "res8_chain" > gpurun_out/r04_t8.log 2>&1; tail -4 gpurun_out/r04_t8.log; grep -n "AssertionError: (" gpurun_out/r04_t8.log | cut -c1-300
