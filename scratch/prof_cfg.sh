#!/bin/bash
# usage: scratch/prof_cfg.sh <acgan|pggan|pix2pix>  -- rocprofv3 kernel stats of scratch/other_configs_bench.py for one configuration
cfg=${1:-pix2pix}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$cfg
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$cfg -o $cfg --output-format csv -- python3 scratch/other_configs_bench.py $cfg > gpurun_out/prof_$cfg.txt 2>&1 < /dev/null
grep -v amdgpu.ids gpurun_out/prof_$cfg.txt | tail -4
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/prof_$cfg/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}%  x{int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
rm -f gpurun_out/prof_$cfg/*kernel_trace.csv
