#!/bin/bash
# usage: bash scratch/ab_set.sh reps steps "args of arm 1" "args of arm 2" ...   -- bench.py arms interleaved on ONE box
# (an arm = extra bench.py arguments, e.g. "--set functional.RES8_CONV=False"; "" = defaults)
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only (no fp16 child, no other configurations)
reps=${1:-2}; steps=${2:-100}; shift 2
for rep in $(seq $reps); do
  for arm in "$@"; do
    python bench.py --no-cpu-baseline --steps $steps --warmup 10 $arm 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('[$arm]', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'])"
  done
done
