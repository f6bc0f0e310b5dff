#!/bin/bash
out=gpurun_out/r5g; mkdir -p $out
export GANK_LIB_NAME=libgank_tune.so
for env in "X=1" "GANK_IMGWG_DBG=1" "GANK_IMGWG_DBG=8" "GANK_IMGWG_DBG=16"; do
  env $env python scratch/r5_imgwg_graph.py 2>/dev/null >> $out/imgwg.log
done
cat $out/imgwg.log
