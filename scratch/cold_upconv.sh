#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for n in 128 320; do for stats in 0 1; do for mode in hot w x wx evict; do
  rm -rf /tmp/cu
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cu -- python3 $GRAFT_REPO_ROOT/scratch/cold_upconv.py $mode $n $stats > /dev/null 2>&1
  echo "== n=$n stats=$stats $mode: $(python3 $GRAFT_REPO_ROOT/scratch/kstat.py $(find /tmp/cu -name '*kernel_stats.csv' | head -1) 'conv_igemm_kernel')"
done; done; done
