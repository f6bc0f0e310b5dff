#!/bin/bash
# usage: bash scratch/ab_env.sh reps steps "ENV=.. ENV=.." "ENV=.." ...   -- bench.py on the TUNING library (libgank_tune.so), one arm per
# environment setting ("X=1" = defaults), interleaved on ONE box
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only (no fp16 child, no other configurations)
reps=${1:-2}; steps=${2:-100}; shift 2
export GANK_LIB_NAME=libgank_tune.so
for rep in $(seq $reps); do
  for arm in "$@"; do
    env $arm python bench.py --no-cpu-baseline --steps $steps --warmup 10 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'])"
  done
done
