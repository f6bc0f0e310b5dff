#!/bin/bash
# correctness of the PP kernel, then A/B timing of G.Block.3.Conv2-shaped convs (old vs new) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GANK_IGEMM_PP=1 timeout -k 5 120 python3 scratch/pp_check.py > gpurun_out/pp_check.log 2>&1 || { tail -20 gpurun_out/pp_check.log; echo "pp_check failed"; exit 1; }
tail -8 gpurun_out/pp_check.log
for pp in 0 1; do
  for n in 128 320; do
    GANK_IGEMM_PP=$pp timeout -k 5 120 python3 scratch/conv_micro.py fprop $n 32 256 256 3 30 || exit 1
  done
done
