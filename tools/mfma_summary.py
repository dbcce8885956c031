#!/usr/bin/env python3
"""profiles/<tag>_mfma_util.csv from the two matrix-core counter passes of tools/profile_round.sh.

    python tools/mfma_summary.py <tag> gpurun_out/prof_<tag>_mfma1 gpurun_out/prof_<tag>_mfma2
"""
import collections
import csv
import glob
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rows = collections.defaultdict(dict)
for d in sys.argv[2:]:
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # gpurun merges runs: the newest
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r"lnasr::(?:\(anonymous namespace\)::)?([a-z_0-9]+)", r["Kernel_Name"])
        agg[m.group(1) if m else r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        for n, v in c.items():
            rows[k][n] = sum(v) / len(v)
cols = ["MfmaUtil", "VALUBusy", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64",
        "SQ_INSTS_VALU_MFMA_MOPS_I8", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]
out = os.path.join(REPO, "profiles", tag + "_mfma_util.csv")
with open(out, "w") as fp:
    fp.write("# rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1` (per-launch averages): MfmaUtil / VALUBusy in percent,\n"
             "# SQ_VALU_MFMA_COEXEC_CYCLES = cycles in which the MFMA unit was busy AND an ordinary vector instruction issued\n")
    fp.write("kernel," + ",".join(cols) + "\n")
    for k in sorted(rows):
        fp.write(k + "," + ",".join("%.6g" % rows[k].get(c, float("nan")) for c in cols) + "\n")
print(open(out).read())
