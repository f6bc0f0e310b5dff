#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/scratch/patch_phase_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size')))
    if 'conv_igemm' not in k[0]: continue
    agg.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    print(f"{k[0]:72s} grid {k[1]:>8s}  x{len(v):3d}  avg {sum(v)/len(v):7.1f} us")
PY
