#!/bin/bash
# ConvMeanPool-form (4x4 stride-2 rows kernel + fold) filter gradient at the generator's UpsampleConv sizes (operands swapped: x := dy at 2H x 2W,
# dy := x at H x W): would the phase form of the UpsampleConv filter gradient (4/9 of the FLOPs) beat the all-taps kernel on the upsampled input?
cd /tmp && export TMPDIR=/tmp
for cfg in "128 32 256 256" "128 16 256 256" "128 8 256 1024"; do
  set -- $cfg
  rm -rf /tmp/cq
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cq -- python3 $GRAFT_REPO_ROOT/scratch/conv_micro.py cpwgrad $1 $2 $3 $4 3 20 > /dev/null 2>&1
  echo "== n=$1 hi-res=$2 'cin'=$3 'cout'=$4"; python3 $GRAFT_REPO_ROOT/scratch/kstat.py $(find /tmp/cq -name '*kernel_stats.csv' | head -1) wgrad
done
