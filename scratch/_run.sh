python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "forward_and_golden or sampling or fixed_noise or headline" > gpurun_out/r04_t17.log 2>&1; tail -3 gpurun_out/r04_t17.log
bash scratch/ab_base.sh 2 100
