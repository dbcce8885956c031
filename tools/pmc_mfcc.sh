#!/bin/bash
# Developer helper (GPU box): SQ counters of the MFCC frame kernel (config 2), two passes of <= 8 SQ counters each.
#   gpurun -- 'bash tools/pmc_mfcc.sh r03'
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
rocprofv3 -L > $O/counters_${tag}.txt 2>&1
rm -rf $O/pmc_${tag}_a $O/pmc_${tag}_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $O/pmc_${tag}_a -- python3 $R/tools/bench_rows.py mfcc > $O/pmc_${tag}_a.log 2>&1
echo "pass a rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $O/pmc_${tag}_b -- python3 $R/tools/bench_rows.py mfcc > $O/pmc_${tag}_b.log 2>&1
echo "pass b rc=$?"
python3 - <<PY
import csv, glob, collections
for p in ("a", "b"):
    for f in glob.glob("$O/pmc_${tag}_%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "mfcc_frame512" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print("   %-28s %16.0f per launch" % (c, v / n[(k, c)]))
PY
