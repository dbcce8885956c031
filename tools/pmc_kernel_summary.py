#!/usr/bin/env python3
"""Condense tools/pmc_kernel.sh output: mean counter value per dispatch for kernels matching a substring.
    python tools/pmc_kernel_summary.py mfcc frame512"""
import csv, glob, os, sys, collections
row, flt = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_%s_*" % row, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt in name:
            short = name.split("(")[0].replace("void lnasr::", "")
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kname, cs in acc.items():
    print(kname)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
