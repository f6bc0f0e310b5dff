#!/bin/bash
# the full GPU suite three times in a row in one box, one log
log=gpurun_out/r04_gpu_suite_3x.txt
: > $log
for i in 1 2 3; do
  echo "=== run $i: python -m pytest tests -m gpu -x -q  ($(date -u +%H:%M:%S))" >> $log
  python -m pytest tests -m gpu -x -q >> $log 2>&1 || { echo "run $i FAILED" >> $log; exit 1; }
done
echo "=== three runs green ($(date -u +%H:%M:%S))" >> $log
