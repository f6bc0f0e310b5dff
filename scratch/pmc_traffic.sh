#!/bin/bash
# HBM traffic of the conv kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots: 3 + 2 do not fit).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_t
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_t -- python3 $GRAFT_REPO_ROOT/scratch/conv_micro.py "$@" > /dev/null 2>&1 < /dev/null || { echo "pass $c failed"; exit 1; }
  python3 - $c <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pmc_t/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    if r['Counter_Name'] == sys.argv[1] and ('conv_' in r['Kernel_Name']) and 'prep' not in r['Kernel_Name']:
        agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(sys.argv[1], k, "launches", len(v), "avg", sum(v) / len(v))
PY
done
