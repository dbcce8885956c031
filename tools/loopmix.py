#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing (development aid).
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -I include -I ln-asr_amd/csrc X.hip -o X.s
    python tools/loopmix.py X.s <mangled-name-prefix>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(re.escape(sys.argv[2]) + r"\w*:", l)][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
for h in [i for i, l in enumerate(body) if "Loop Header" in l]:
    lab = body[h].split(":")[0].strip()
    ends = [i for i, l in enumerate(body) if ("s_cbranch" in l or "s_branch" in l) and l.split()[-1] == lab and i > h]
    if not ends:
        continue
    seg = body[h:ends[-1] + 1]
    c = Counter()
    for l in seg:
        l = l.strip()
        if not l or l[0] in ";.":
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif "_dpp" in op: c["valu_dpp"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op == "s_barrier": c["barrier"] += 1
        elif op == "s_waitcnt": c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["vmem"] += 1
    print(lab, "lines", len(seg), dict(c))
