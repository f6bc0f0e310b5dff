#!/bin/bash
# the data-parallel code path at world size 1 (bench.py --force-dp), several fresh processes: exit code, wall time, last stderr lines
for i in 1 2 3 4 5 6; do
  t0=$(date +%s)
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --force-dp > gpurun_out/fdp_$i.out 2> gpurun_out/fdp_$i.err
  rc=$?
  t1=$(date +%s)
  echo "run $i: rc=$rc wall=$((t1 - t0))s json=$(grep -c '"metric"' gpurun_out/fdp_$i.out) :: $(grep -v amdgpu.ids gpurun_out/fdp_$i.err | grep -v hostname | tail -2 | cut -c1-300)"
  [ $rc -ne 0 ] && [ $rc -ne 124 ] && break
done
