#!/bin/bash
# Developer helper (runs on the GPU box): hardware counters of one bench_rows.py row, one pass per counter group.
#   gpurun -- 'bash tools/pmc_kernel.sh mfcc ["CTR1 CTR2" ...]'      -> gpurun_out/pmc_<row>_<k>/ (csv)
row=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
k=0
if [ -n "$2" ]; then groups=("${@:2}"); else groups=(
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM"
    "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
    "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY"
    "VALUBusy SALUBusy LDSBankConflict MemUnitStalled"); fi
for grp in "${groups[@]}"; do
    k=$((k + 1))
    rm -rf $O/pmc_${row}_$k
    rocprofv3 --pmc $grp -f csv -d $O/pmc_${row}_$k -- python3 $R/tools/bench_rows.py $row > /dev/null 2> $O/pmc_${row}_$k.err || echo "group $k failed"
done
echo pmc done
