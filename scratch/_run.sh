python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "convpool3x3_resident" > gpurun_out/r04_t3.log 2>&1; tail -3 gpurun_out/r04_t3.log
echo "== k2 on"; GANK_LIB_NAME=libgank_tune.so python scratch/cpool_bench.py
echo "== k2 off"; GANK_LIB_NAME=libgank_tune.so GANK_CPOOL_K2=0 python scratch/cpool_bench.py
bash scratch/ab_base.sh 2 100
