#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_c -- python3 $GRAFT_REPO_ROOT/scratch/conv_micro.py "$@" > /dev/null 2>&1 < /dev/null || { echo "failed"; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_c/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'conv_' in r['Kernel_Name'] and 'prep' not in r['Kernel_Name']]
cols = rows[0].keys()
vals = [float(r['Counter_Value']) for r in rows]
dur = None
if 'Start_Timestamp' in cols:
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows]
print("launches", len(vals), "GRBM_GUI_ACTIVE avg", sum(vals) / len(vals))
if dur:
    d = sum(dur) / len(dur)
    print("avg duration ns", d, "effective clock GHz", sum(vals) / len(vals) / 8 / d)
else:
    print(list(cols))
PY
