#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_t -o t --output-format csv -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_t/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if any(k in n for k in ('cbn_', 'sn_', 'unpool', 'pool2x2', 'ew_kernel', 'concat', 'prep_batch', 'adam', 'copy_bytes', 'Fill')):
        key = (n[:34], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']))
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
iters = 11.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(f"{k[0]:34s} blocks {k[1]:6d}x{k[2]:<3d} calls/it {len(v)/iters:5.1f} avg_us {sum(v)/len(v):7.1f} ms/it {sum(v)/1e3/iters:6.3f}")
PY
rm -rf gpurun_out/prof_t
