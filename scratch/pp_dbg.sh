#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 256 1280; do
  echo "dbg $d"
  GANK_PP_DBG=$d GANK_IGEMM_PP=1 timeout -k 5 120 python3 scratch/conv_micro.py fprop 128 32 256 256 3 30 || exit 1
done
