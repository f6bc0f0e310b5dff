#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for burst in 0 20; do
  rm -rf /tmp/cb
  rocprofv3 --kernel-trace --output-format csv -d /tmp/cb -- python3 $GRAFT_REPO_ROOT/scratch/clock_after_burst.py $burst > /dev/null 2>&1
  python3 - $burst <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/cb/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
img = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'img16' in r['Kernel_Name']]
pp = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'conv_igemm_pp' in r['Kernel_Name']]
print("burst", sys.argv[1], "img16 launches (us), last 3 graph replays:", [round(v, 1) for v in img[-18:]])
if pp: print("   pp kernel first / last of a burst:", [round(v, 1) for v in pp[-20:]][:3], [round(v, 1) for v in pp[-3:]])
PY
done
