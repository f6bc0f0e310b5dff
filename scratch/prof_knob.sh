#!/bin/bash
# usage: prof_knob.sh <kernel substring> "ENV=.." "ENV=.." ...  -- in-step kernel durations (rocprofv3 --kernel-trace --stats of bench.py on the tuning library) per setting
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
export GANK_LIB_NAME=libgank_tune.so
for arm in "$@"; do
  rm -rf /tmp/pk; export $arm
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
  echo "== $arm"; python3 $GRAFT_REPO_ROOT/scratch/kstat.py $(find /tmp/pk -name '*kernel_stats.csv' | head -1) "$pat"
done
