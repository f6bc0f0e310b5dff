#!/bin/bash
# usage: r5_ab_set.sh reps "args of arm 1" "args of arm 2" ...   -- bench.py arms (extra arguments, e.g. --set module.NAME=False) interleaved on one box
export GANK_BENCH_EXTRAS=0
reps=$1; shift
for rep in $(seq $reps); do
  for arm in "$@"; do
    python bench.py --no-cpu-baseline --steps 100 --warmup 10 $arm 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'], d['config']['finite'])"
  done
done
