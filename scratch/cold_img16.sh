#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for half in 0 1; do for mode in hot w x wx evict; do
  rm -rf /tmp/ci; 
  GANK_LIB_NAME=libgank_tune.so GANK_IMG16_HALF=$half rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ci -- python3 $GRAFT_REPO_ROOT/scratch/cold_img16.py $mode > /dev/null 2>&1
  echo "== half=$half $mode"; python3 $GRAFT_REPO_ROOT/scratch/kstat.py $(find /tmp/ci -name '*kernel_stats.csv' | head -1) img16
done; done
