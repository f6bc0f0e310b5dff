python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv_fprop or cbn_relu_fused" > gpurun_out/r04_t23.log 2>&1; tail -2 gpurun_out/r04_t23.log
python scratch/bench_few.py
