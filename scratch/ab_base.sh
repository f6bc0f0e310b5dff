#!/bin/bash
# usage: scratch/ab_base.sh [reps] [steps] ["ENV=.. ENV=.." ...]   -- the round-4 tree (_baseline_r04, a git worktree of 282d0e9 with its own
# libgank.so, not tracked) against the current tree, bench.py interleaved on ONE box; extra arguments: environment settings of further
# arms of the current tree
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only (no fp16 child, no other configurations)
reps=${1:-2}; steps=${2:-100}; shift 2 2>/dev/null
root=$(pwd)
one() {   # dir, label, env...
  local dir=$1 label=$2; shift 2
  ( cd $dir && env "$@" python bench.py --no-cpu-baseline --steps $steps --warmup 10 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$label', d['value'], 'img/s', d['ms_per_step'], 'ms', 'median', d['median_ms_per_step_hip_events'])" )
}
for rep in $(seq $reps); do
  [ -d $root/_baseline_r04 ] && one $root/_baseline_r04 "r04-baseline" X=1
  one $root "current" X=1
  for kv in "$@"; do one $root "current[$kv]" $kv; done
done
