python -m pytest tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r04_t28.log 2>&1; tail -3 gpurun_out/r04_t28.log
bash scratch/ab_base.sh 2 100
