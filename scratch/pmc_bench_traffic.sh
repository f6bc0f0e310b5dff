#!/bin/bash
# HBM traffic of the dominant kernel family over bench.py's workload: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc
# passes (MI355X_MICROARCH.md: TCC has 4 slots, the two need 3 + 2), units KB, FETCH_SIZE doubled on gfx950.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_b_$c
  timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_b_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1 < /dev/null || { echo "pass $c failed"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, json, collections
res = {}
fam = lambda n: ('conv_igemm' in n or 'conv_narrow' in n)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f'/tmp/pmc_b_{c}/**/*counter_collection.csv', recursive=True)[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == c and fam(r['Kernel_Name'])]
    res[c] = {"launches": len(vals), "avg_kb_per_launch": sum(vals) / max(len(vals), 1)}
fetch = 2.0 * 1024.0 * res["FETCH_SIZE"]["avg_kb_per_launch"]      # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
write = 1024.0 * res["WRITE_SIZE"]["avg_kb_per_launch"]
res["bytes_per_launch"] = fetch + write
res["fetch_bytes_per_launch_corrected"] = fetch
res["write_bytes_per_launch"] = write
res["kernel_family"] = "conv_igemm_* + conv_narrow_in (fprop + dgrad), all launches of bench.py --steps 3 --warmup 2"
res["method"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB -> bytes; FETCH_SIZE x2 (gfx950)"
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res))
PY
