#!/bin/bash
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
for grp in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY TCC_BUSY_avr"; do
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/scratch/conv_micro.py "$@" 6 > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(f'/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv', recursive=True)
if not f: print("no counter file"); sys.exit()
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'conv_' in r['Kernel_Name'] and 'prep' not in r['Kernel_Name']:
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
