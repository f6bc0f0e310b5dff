#!/bin/bash
# usage: scratch/ab_env.sh "VAR=a" "VAR=b" ... -- bench.py under each environment setting, interleaved twice
for rep in 1 2; do
  for kv in "$@"; do
    env $kv python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$kv', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'])"
  done
done
