#!/bin/bash
# correctness of the PP kernel forms, then A/B timing against the older kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GANK_IGEMM_PP=2 timeout -k 5 150 python3 scratch/pp_check.py > gpurun_out/pp_check.log 2>&1 || { tail -30 gpurun_out/pp_check.log; echo "pp_check failed"; exit 1; }
tail -3 gpurun_out/pp_check.log
for pp in 0 2; do
  echo "GANK_IGEMM_PP=$pp"
  GANK_IGEMM_PP=$pp timeout -k 5 120 python3 scratch/conv_micro.py fprop 128 32 256 256 3 30 || exit 1
  GANK_IGEMM_PP=$pp timeout -k 5 120 python3 scratch/conv_micro.py fprop 320 32 256 256 3 30 || exit 1
  GANK_IGEMM_PP=$pp timeout -k 5 120 python3 scratch/conv_micro.py fprop 128 16 256 256 3 30 || exit 1
  GANK_IGEMM_PP=$pp timeout -k 5 120 python3 scratch/conv_micro.py fprop 320 16 256 256 3 30 || exit 1
  GANK_IGEMM_PP=$pp timeout -k 5 120 python3 scratch/conv_micro.py upfprop 128 32 256 256 3 30 || exit 1
done
