#!/bin/bash
# LDS bank-conflict share per kernel over bench.py's workload: SQ_LDS_BANK_CONFLICT (extra cycles) / SQ_LDS_IDX_ACTIVE (all LDS-array
# cycles), one --pmc pass; plus SQ_LDS_UNALIGNED_STALL
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
timeout -k 10 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL --output-format csv -d /tmp/pmc_lds -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1 < /dev/null || { echo "pass failed"; exit 1; }
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/pmc_lds/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:70]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": cnt[k] += 1
rows = sorted(per.items(), key=lambda kv: -kv[1]["SQ_LDS_BANK_CONFLICT"])
for k, d in rows[:30]:
    a = d["SQ_LDS_IDX_ACTIVE"]
    print(f'{k:70s} n={cnt[k]:4d} conflict {d["SQ_LDS_BANK_CONFLICT"]:14.0f} active {a:14.0f} share {d["SQ_LDS_BANK_CONFLICT"]/a if a else 0:5.2f} unaligned {d["SQ_LDS_UNALIGNED_STALL"]:12.0f}')
PY
