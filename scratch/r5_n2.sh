#!/bin/bash
out=gpurun_out/r5m; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "image_resident_16x16 or img16" > $out/k.log 2>&1; echo "rc=$?" >> $out/k.log; tail -5 $out/k.log
python -m pytest tests/test_timed_path_gpu.py -x -q -m gpu > $out/m.log 2>&1; echo "rc=$?" >> $out/m.log; tail -3 $out/m.log
for rep in 1 2; do
for arm in "--set common.resnet_block.FUSE_NORM_INTO_CONV2=False" "--set common.resnet_block.FUSE_NORM_INTO_CONV2=True"; do
  python bench.py --no-cpu-baseline --steps 100 --warmup 10 $arm 2>/dev/null > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'], d['config']['finite'])" | tee -a $out/ab.log
done; done
( time python bench.py --steps 200 --warmup 20 > $out/bench_full.json 2> $out/bench_full.err ) 2> $out/bench_time.txt
tail -3 $out/bench_time.txt; python -c "
import json; d=json.load(open('$out/bench_full.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('fp16'))); print(json.dumps(d.get('other_configs'), indent=1)); print(d['cpu_baseline']['value'])"
