#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # mfma16 args...
  rm -rf gpurun_out/sw
  GANK_IGEMM_MFMA16=$1 timeout -k 5 90 rocprofv3 --kernel-trace --stats -d gpurun_out/sw -o sw --output-format csv -- python3 scratch/conv_micro.py "${@:2}" > /dev/null 2>&1 < /dev/null || { echo "run failed"; exit 1; }
  echo "mfma16 $1 [${@:2}]:"; python3 scratch/kstat.py gpurun_out/sw/sw_kernel_stats.csv conv_igemm
}
for f in 0 1 0 1; do
  run $f fprop 128 32 256 256 3 20
done
for f in 0 1; do
  run $f fprop 320 32 256 256 3 10
  run $f fprop 128 16 256 256 3 30
done
rm -rf gpurun_out/sw
