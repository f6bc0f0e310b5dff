#!/bin/bash
# round 5: XCD-aware block order of the filter-gradient and resident kernels -- micro timings, interleaved whole-step A/B on the TUNING library
out=gpurun_out/r5b; mkdir -p $out
python -m pytest tests/test_timed_path_gpu.py -x -q -m gpu -s > $out/timed.log 2>&1; echo "timed rc=$?" >> $out/timed.log
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad or cpool or img16 or res8 or convpool or upconv or filter_gradient" > $out/kernels.log 2>&1; echo "kernels rc=$?" >> $out/kernels.log
for x in 0 1; do
  echo "== GANK_WGRAD_XCD=$x" >> $out/micro.log
  GANK_LIB_NAME=libgank_tune.so GANK_WGRAD_XCD=$x python scratch/bench_critic_wgrad.py rows cpool taps >> $out/micro.log 2>&1
done
bash scratch/ab_env.sh 2 100 "GANK_WGRAD_XCD=0 GANK_RESIDENT_XCD=0" "GANK_WGRAD_XCD=1 GANK_RESIDENT_XCD=0" "GANK_WGRAD_XCD=0 GANK_RESIDENT_XCD=1" "GANK_WGRAD_XCD=1 GANK_RESIDENT_XCD=1" > $out/ab.log 2>&1
tail -3 $out/timed.log; tail -3 $out/kernels.log; cat $out/micro.log; cat $out/ab.log
