#!/bin/bash
# usage: scratch/ab.sh ENVVAR v1 v2 [reps]   -- interleaved A/B of bench.py under two values of an environment knob (one box, one call)
var=$1; a=$2; b=$3; reps=${4:-2}
for i in $(seq $reps); do
  for v in $a $b; do
    env $var=$v python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$var=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'])"
  done
done
