#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name target minsteps args...
  rm -rf gpurun_out/sw
  GANK_WGRAD_SPLIT_TARGET=$1 GANK_WGRAD_MIN_STEPS=$2 timeout -k 5 90 rocprofv3 --kernel-trace --stats -d gpurun_out/sw -o sw --output-format csv -- python3 scratch/conv_micro.py "${@:3}" > /dev/null 2>&1 < /dev/null || { echo "run failed"; exit 1; }
  echo "target $1 minsteps $2 [${@:3}]:"; python3 scratch/kstat.py gpurun_out/sw/sw_kernel_stats.csv wgrad
}
for cfg in "768 4" "384 8" "256 8" "128 8" "64 8"; do set -- $cfg
  run $1 $2 cpwgrad 128 32 128 128 3 30
  run $1 $2 cpwgrad 128 16 256 128 3 30
done
rm -rf gpurun_out/sw
