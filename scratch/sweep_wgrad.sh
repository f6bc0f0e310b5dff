#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # minm args...
  rm -rf gpurun_out/sw
  GANK_WGRAD_TAPS_MINM=$1 timeout -k 5 90 rocprofv3 --kernel-trace --stats -d gpurun_out/sw -o sw --output-format csv -- python3 scratch/conv_micro.py "${@:2}" > /dev/null 2>&1 < /dev/null || { echo "run failed"; exit 1; }
  echo "taps_minm $1 [${@:2}]:"; python3 scratch/kstat.py gpurun_out/sw/sw_kernel_stats.csv wgrad
}
run 16384 wgrad_relu 128 8 128 128 3 50
run 4096 wgrad_relu 128 8 128 128 3 50
rm -rf gpurun_out/sw
