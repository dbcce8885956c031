#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small files committed under profiles/.

    python tools/prof_summary.py <tag> <trace_dir> [<fetch_dir> <write_dir>]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim) and, when the
two --pmc passes are given, profiles/<tag>_pmc.csv plus profiles/pmc_traffic.json (HBM bytes per launch:
(2 * FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE reads exactly half of a coalesced streaming read on
gfx950 (MI355X_MICROARCH.md, HBM); checked here against the known sizes of obs / b / v)."""
import collections
import csv
import datetime
import glob
import hashlib
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, trace = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
ks = max(glob.glob(os.path.join(trace, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # gpurun merges runs: the newest
shutil.copy(ks, os.path.join(REPO, "profiles", tag + "_kernel_stats.csv"))
print(open(ks).read()[:1500])


def short(name):
    m = re.search(r"lnasr::(?:\(anonymous namespace\)::)?([a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


if len(sys.argv) >= 5:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[3:5]:
        f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)   # gpurun merges runs: the newest
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    with open(os.path.join(REPO, "profiles", tag + "_pmc.csv"), "w") as fp:
        fp.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
        for k, c in sorted(agg.items()):
            fe = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0])))
            wr = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0])))
            hbm = (2 * fe + wr) * 1024
            fp.write("%s,%d,%.1f,%.1f,%.0f\n" % (k, len(c.get("FETCH_SIZE", [])), fe, wr, hbm))
            out[k] = {"fetch_kib": fe, "write_kib": wr, "hbm_bytes_per_launch": hbm}
    key_map = {"emit_diag_mfma": "emit_diag_mfma", "emit_diag_i8": "emit_diag_i8", "slice_features_kernel": "slice_features",
               "viterbi_reg": "viterbi_reg_n128", "viterbi_reg1b": "viterbi_reg1b_n128",
               "backtrace_lds_kernel": "backtrace_lds"}
    traffic = {v: out[k] for k, v in key_map.items() if k in out}
    # which source the counted kernel was built from: bench.py prints a traffic figure only while this still matches
    src_of = {"emit_diag_mfma": "gmm.hip", "emit_diag_i8": "gmm_i8.hip", "slice_features": "gmm_i8.hip",
              "viterbi_reg_n128": "viterbi.hip", "viterbi_reg1b_n128": "viterbi.hip", "backtrace_lds": "backtrace.hip"}
    csrc = os.path.join(REPO, "ln-asr_amd", "csrc")
    for name, ent in traffic.items():
        h = hashlib.sha256()
        for f in (src_of[name], "devmath.h"):                 # the kernel's source and the device math it inlines
            h.update(open(os.path.join(csrc, f), "rb").read())
        ent.update(tag=tag, date=datetime.date.today().isoformat(), source="ln-asr_amd/csrc/" + src_of[name],
                   source_sha16=h.hexdigest()[:16])
    traffic["_source"] = "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py (U=1024, T=1000)" % tag
    json.dump(traffic, open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(REPO, "profiles", tag + "_pmc.csv")).read())
