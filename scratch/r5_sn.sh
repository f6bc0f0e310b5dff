#!/bin/bash
out=gpurun_out/r5i; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "spectral_norm or adam" > $out/k.log 2>&1; echo "rc=$?" >> $out/k.log; tail -4 $out/k.log
for rep in 1 2; do
for arm in "--set SNGAN.gan_cifar_resnet.FUSE_SN_TAIL=False" "--set SNGAN.gan_cifar_resnet.FUSE_SN_TAIL=True"; do
  python bench.py --no-cpu-baseline --steps 100 --warmup 10 $arm 2>/dev/null > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'], d['config']['finite'])" | tee -a $out/ab.log
done; done
bash scratch/measure.sh r5i > $out/measure.log 2>&1
grep -n "sn_\|adam\|critic_feed" gpurun_out/seq_r5i.txt | head -12
