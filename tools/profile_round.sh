#!/bin/bash
# Runs on the GPU box (through gpurun): the driver-facing bench, then the rocprofv3 passes whose summaries
# tools/prof_summary.py condenses into profiles/<tag>_*.  Counters are collected in their own passes.
#   gpurun -- 'bash tools/profile_round.sh r01d'
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $O/bench_$tag.json 2> $O/bench_$tag.err
tail -c 600 $O/bench_$tag.json
# one GPU holding ALL 8192 utterances of configs[3] (2.5 GB of observations, 8.4 GB each of b and v): throughput is
# flat in the batch size, the 1024-utterance shard is not a favourable slice
python3 $R/bench.py --steps 5 --warmup 1 --utts 8192 --no-cpu-baseline --no-extras > $O/bench_${tag}_u8192.json 2> $O/bench_${tag}_u8192.err || echo "8192-utterance run failed"
tail -c 300 $O/bench_${tag}_u8192.json
rm -rf $O/prof_${tag}_trace $O/prof_${tag}_fetch $O/prof_${tag}_write
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_${tag}_trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_${tag}_trace.json 2> $O/prof_${tag}_trace.err
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/prof_${tag}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/prof_${tag}_fetch.err
rocprofv3 --pmc WRITE_SIZE -f csv -d $O/prof_${tag}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/prof_${tag}_write.err

# matrix-core utilisation and MFMA / VALU co-execution (tools/mfma_summary.py)
rm -rf $O/prof_${tag}_mfma1 $O/prof_${tag}_mfma2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 -f csv -d $O/prof_${tag}_mfma1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/prof_${tag}_mfma1.err
rocprofv3 --pmc MfmaUtil VALUBusy -f csv -d $O/prof_${tag}_mfma2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/prof_${tag}_mfma2.err

# the other rows of SURVEY 8d (MFCC, single GMM, bank, forward / backward, Baum-Welch statistics, full covariance)
rm -rf $O/prof_${tag}_rows
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_${tag}_rows -- python3 $R/tools/bench_rows.py all > $O/rows_${tag}.log 2> $O/prof_${tag}_rows.err
echo rows done
