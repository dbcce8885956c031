#!/bin/bash
# Developer helper (GPU box): SQ counters of the kernels of one tools/bench_rows.py row, two passes of <= 8 counters.
#   gpurun -- 'bash tools/pmc_rows.sh <tag> <row> <kernel-substring>'      e.g.  r03c bw mix_mfma
tag=${1:-r03}
row=${2:-bw}
pat=${3:-mix_mfma}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
rm -rf $O/pmc_${tag}_a $O/pmc_${tag}_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -f csv -d $O/pmc_${tag}_a -- python3 $R/tools/bench_rows.py $row > $O/pmc_${tag}_a.log 2>&1
echo "pass a rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -f csv -d $O/pmc_${tag}_b -- python3 $R/tools/bench_rows.py $row > $O/pmc_${tag}_b.log 2>&1
echo "pass b rc=$?"
python3 - <<PY
import csv, glob, collections
for p in ("a", "b"):
    for f in glob.glob("$O/pmc_${tag}_%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            if "$pat" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print("   %-32s %18.0f per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
PY
