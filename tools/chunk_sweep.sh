#!/bin/bash
# Developer helper (GPU box): decode step time against LNASR_DECODE_CHUNKS (chunked two-stream decode).
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in 1 2 4 8; do
  LNASR_DECODE_CHUNKS=$c python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/b_$c.json 2>/tmp/b_$c.err || { echo "chunks=$c FAILED"; tail -3 /tmp/b_$c.err; continue; }
  python3 - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.load(open("/tmp/b_%s.json"%c))
print("chunks=%s ms_per_step=%.3f"%(c,d["ms_per_step"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
done
