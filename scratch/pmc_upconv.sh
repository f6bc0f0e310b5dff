#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of isolated UpsampleConv (phase) and plain conv launches
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_u_$c
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_u_$c -- python3 $GRAFT_REPO_ROOT/scratch/upconv_once.py > /dev/null 2>&1 < /dev/null || { echo "pass $c failed"; exit 1; }
  python3 - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob(f'/tmp/pmc_u_{c}/**/*counter_collection.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c and 'conv_igemm' in r["Kernel_Name"]:
        print(c, r["Kernel_Name"][:60], round(float(r["Counter_Value"]) * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6, 1), 'MB')
PY
done
