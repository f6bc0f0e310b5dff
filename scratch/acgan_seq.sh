#!/bin/bash
# usage: acgan_seq.sh [batch]  -> gpurun_out/acgan_seq_<batch>.txt
bs=${1:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/aq
rocprofv3 --kernel-trace --output-format csv -d /tmp/aq -- python3 $GRAFT_REPO_ROOT/scratch/acgan_seq.py $bs > /dev/null 2>&1
python3 - $bs <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/acgan_seq_$bs.txt
import csv, glob, re, sys, collections
f = glob.glob('/tmp/aq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    n = re.sub(r'^void ', '', r['Kernel_Name']); return re.sub(r'\(.*', '', n)[:70]
# the last two steps are graph replays: a step = 6 graph launches; find step boundaries by the largest gaps
t = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
n = len(rows)
# take the last 1/8 of the trace (8 iterations, the last ones replayed) 
lo = n - n // 8 if n > 800 else 0
# refine: replayed steps have identical lengths; use the last step = rows after the last gap > 200 us followed by >= 500 rows
seg = rows[lo:]
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
print(f"tail of the trace: {len(seg)} launches, kernel time {tot:.0f} us, wall span {span:.0f} us")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    a = agg[nm(r)]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {us:9.1f} us  x{c:5d}  avg {us / c:7.1f}  {k}")
print("---- sequence")
prev = None
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {nm(r)}")
    prev = e
PY
head -60 $GRAFT_REPO_ROOT/gpurun_out/acgan_seq_$bs.txt
