#!/bin/bash
# the data-parallel code path on ONE GPU (world-size-1 RCCL group) against the plain single-GPU path, interleaved
for rep in 1 2; do
  for a in "" "--force-dp" "--force-dp --no-capture-collectives" "--force-dp --grad-wire bf16"; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10 $a 2>/dev/null > /tmp/dp.out
    python - "$a" <<'PY'
import json, sys
d = [json.loads(l) for l in open('/tmp/dp.out') if l.startswith('{')][-1]
print(f"[{sys.argv[1] or 'plain'}]", d["value"], "img/s", d["ms_per_step"], "ms  collectives_in_graphs =", d["config"]["collectives_in_graphs"])
PY
  done
done
