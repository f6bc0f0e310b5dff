#!/bin/bash
# G.Block.1.Conv1 alone (hot, graph replay) over batch sizes: how the 128 x 128-tile generic kernel fills 256 CUs
cd /tmp && export TMPDIR=/tmp
for n in 192 256 320 384 512 640; do
  rm -rf /tmp/cu
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cu -- python3 $GRAFT_REPO_ROOT/scratch/cold_upconv.py hot $n 1 > /dev/null 2>&1
  echo "== n=$n: $(python3 $GRAFT_REPO_ROOT/scratch/kstat.py $(find /tmp/cu -name '*kernel_stats.csv' | head -1) 'conv_igemm_kernel')"
done
