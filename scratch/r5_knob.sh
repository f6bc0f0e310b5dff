#!/bin/bash
# usage: r5_knob.sh "pat1|pat2" "ENV=.." ...  -- in-step kernel durations per setting of the TUNING library (regex on kernel names)
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only (no fp16 child, no other configurations)
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
export GANK_LIB_NAME=libgank_tune.so
for arm in "$@"; do
  rm -rf /tmp/pk
  env $arm rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /tmp/pk.out 2> /tmp/pk.err || tail -5 /tmp/pk.err
  echo "== $arm"
  python3 - "$pat" <<'PY'
import csv, glob, re, sys
f = glob.glob('/tmp/pk/**/*kernel_stats.csv', recursive=True)
if not f:
    print("  (no stats)"); sys.exit()
for r in csv.DictReader(open(f[0])):
    if re.search(sys.argv[1], r['Name']):
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
