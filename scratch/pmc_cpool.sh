#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of isolated resident ConvMeanPool launches (separate passes; FETCH_SIZE doubled on gfx950)
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_c_$c
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_c_$c -- python3 $GRAFT_REPO_ROOT/scratch/cpool_once.py > /dev/null 2>&1 < /dev/null || { echo "pass $c failed"; exit 1; }
  python3 - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob(f'/tmp/pmc_c_{c}/**/*counter_collection.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c and 'cpool_res' in r["Kernel_Name"]:
        print(c, r["Kernel_Name"][:60], 'grid', r.get("Grid_Size", "?"), round(float(r["Counter_Value"]) * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6, 1), 'MB')
PY
done
