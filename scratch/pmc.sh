#!/bin/bash
# usage: pmc.sh <tag> <conv_micro args...>
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM"; do
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/scratch/conv_micro.py "$@" 8 > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(f'/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv', recursive=True)
if not f: print("no counter file"); sys.exit()
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if ('conv_' in r['Kernel_Name'] or 'conv3x3' in r['Kernel_Name']) and 'prep' not in r['Kernel_Name']:
        agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
