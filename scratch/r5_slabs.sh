#!/bin/bash
out=gpurun_out/r5j; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "batched_equals_separate or image_convs_filter_gradient or spectral_norm" > $out/k.log 2>&1; echo "rc=$?" >> $out/k.log; tail -6 $out/k.log
python -m pytest tests/test_timed_path_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "not synthetic_class and not short_training" > $out/m.log 2>&1; echo "rc=$?" >> $out/m.log; tail -5 $out/m.log
for rep in 1 2; do
for arm in "--set functional.SLAB_WGRADS=False" "--set functional.SLAB_WGRADS=True"; do
  python bench.py --no-cpu-baseline --steps 100 --warmup 10 $arm 2>/dev/null > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'], d['config']['finite'])" | tee -a $out/ab.log
done; done
bash scratch/measure.sh r5j > $out/measure.log 2>&1
grep -n "imgwg\|rows_kernel<1, 2, false>\|sum_slabs" gpurun_out/seq_r5j.txt | head -6
