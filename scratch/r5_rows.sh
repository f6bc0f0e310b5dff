#!/bin/bash
out=gpurun_out/r5c; mkdir -p $out
export GANK_LIB_NAME=libgank_tune.so
for env in "X=1" "GANK_WGRAD_DBG=1" "GANK_WGRAD_DBG=2" "GANK_WGRAD_DBG=3" "GANK_WGRAD_ROWS_TARGET=128" "GANK_WGRAD_ROWS_TARGET=128 GANK_WGRAD_DBG=1" "GANK_WGRAD_ROWS_TARGET=512" "GANK_WGRAD_ROWS_TARGET=512 GANK_WGRAD_DBG=1" "GANK_WGRAD_ROWS_PF=3" "GANK_WGRAD_ROWS_PF=3 GANK_WGRAD_DBG=1"; do
  env $env python scratch/r5_rows_graph.py 2>/dev/null >> $out/rows.log
done
cat $out/rows.log
