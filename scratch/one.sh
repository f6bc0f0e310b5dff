#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sw
timeout -k 5 90 rocprofv3 --kernel-trace --stats -d gpurun_out/sw -o sw --output-format csv -- python3 scratch/conv_micro.py "$@" > /dev/null 2>&1 < /dev/null || { echo "run failed"; exit 1; }
python3 scratch/kstat.py gpurun_out/sw/sw_kernel_stats.csv conv_
rm -rf gpurun_out/sw
