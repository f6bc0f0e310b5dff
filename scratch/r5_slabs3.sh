#!/bin/bash
out=gpurun_out/r5l; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "split_slabs or batched_equals or conv_wgrad or fold_left or convpool" > $out/k.log 2>&1; echo "rc=$?" >> $out/k.log; tail -6 $out/k.log
python -m pytest tests/test_timed_path_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "not synthetic_class and not short_training" > $out/m.log 2>&1; echo "rc=$?" >> $out/m.log; tail -5 $out/m.log
bash scratch/ab_base.sh 3 100 > $out/ab.log 2>&1; cat $out/ab.log
bash scratch/measure.sh r5l > $out/measure.log 2>&1
grep -n "sum_slabs\|fold\|reduce_slabs" gpurun_out/seq_r5l.txt | head -12
