#!/bin/bash
# HBM traffic of the dominant kernel family over bench.py's workload: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc
# passes (MI355X_MICROARCH.md: TCC has 4 slots, the two need 3 + 2), units KB, FETCH_SIZE doubled on gfx950.
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_b_$c
  timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_b_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1 < /dev/null || { echo "pass $c failed"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, json, collections, re
def sym(n):      # rocprofv3 prints "void name<...>(ArgTypes)": keep name<...> as libgank's launchers record it
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\((\(anonymous namespace\)::)?(IgemmArgs|WgradArgs|ResFwdArgs|ResBwdArgs|CpFwdArgs|CpBwdArgs|CpBwdWgArgs|G8Args|I16Args|NarrowWgArgs, NarrowWgArgs, int)\)$", "", n)
    return re.sub(r"\(.*\)$", "", n)            # plain kernels: drop the argument list
per = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f'/tmp/pmc_b_{c}/**/*counter_collection.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and any(t in r["Kernel_Name"] for t in ("conv_", "conv3x3", "res8_", "cpool_res", "wgrad_", "sum_slabs")) and "prep" not in r["Kernel_Name"]:
            per[sym(r['Kernel_Name'])][c].append(float(r['Counter_Value']))
res = {"per_kernel": {}}
for k, d in per.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    fetch = 2.0 * 1024.0 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])     # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    write = 1024.0 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    res["per_kernel"][k] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_per_launch_corrected": fetch,
                            "write_bytes_per_launch": write, "bytes_per_launch": fetch + write}
res["workload"] = "every conv kernel launch of bench.py --no-cpu-baseline --steps 3 --warmup 2"
res["method"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB -> bytes; FETCH_SIZE x2 (gfx950)"
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(res["per_kernel"].items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(k, v)
PY
