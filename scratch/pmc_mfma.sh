#!/bin/bash
# matrix-pipe share per kernel over bench.py's workload: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) against
# GRBM_GUI_ACTIVE (sum over the 8 XCDs) -> busy / (active / 8 * 1024 SIMDs); effective clock = active / 8 / duration needs the trace
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mf
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/pmc_mf -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1 < /dev/null || { echo "pass failed"; exit 1; }
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/pmc_mf/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", ""))[:64]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
tot = sum(d["GRBM_GUI_ACTIVE"] for d in per.values())
print(f'{"kernel":64s} {"n":>4s} {"active share":>12s} {"MFMA busy / SIMD-cycles":>24s}')
for k, d in sorted(per.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:45]:
    a = d["GRBM_GUI_ACTIVE"] / 8.0
    print(f'{k:64s} {cnt[k]:4d} {d["GRBM_GUI_ACTIVE"]/tot:12.3f} {d["SQ_VALU_MFMA_BUSY_CYCLES"]/(a*1024) if a else 0:24.3f}')
PY
