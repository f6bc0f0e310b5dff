#!/bin/bash
out=gpurun_out/r5d; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "image_convs_filter_gradient or convpool3x3_resident" > $out/k.log 2>&1; echo "rc=$?" >> $out/k.log
tail -15 $out/k.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "image_side_filter or d_and_g_gradients or headline" > $out/m.log 2>&1; echo "rc=$?" >> $out/m.log
tail -5 $out/m.log; grep "fused vs" $out/m.log
for rep in 1 2; do
for arm in "--set functional.FUSE_IMAGE_WGRAD=False" "--set functional.FUSE_IMAGE_WGRAD=True"; do
  python bench.py --no-cpu-baseline --steps 100 --warmup 10 $arm 2>/dev/null > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'])" | tee -a $out/ab.log
done; done
