#!/bin/bash
# usage: native_kernels.sh <acgan32|pggan|pix2pix>  -- framework (at::native, rocBLAS Cijk_*) kernels left in one configuration's profile
cfg=${1:-pix2pix}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/nk
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/nk -o $cfg --output-format csv -- python3 scratch/other_configs_bench.py $cfg > /tmp/nk.txt 2>&1 < /dev/null
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/nk/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
nat = [r for r in rows if 'at::native' in r['Name'] or 'Cijk' in r['Name'] or 'rocclr' in r['Name']]
print(f"$cfg: total kernel time {tot/1e6:.1f} ms, {sum(int(r['Calls']) for r in rows)} launches; framework kernels: {sum(float(r['TotalDurationNs']) for r in nat)/1e6:.2f} ms, {sum(int(r['Calls']) for r in nat)} launches")
for r in nat:
    print(f"  {float(r['TotalDurationNs'])/1e6:8.2f} ms  x{int(r['Calls']):5d}  {r['Name'][:150]}")
PY
