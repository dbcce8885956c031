#!/bin/bash
# on the GPU box: kernel times of tools/i8_try.py big
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_i8
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_i8 -- python3 $R/tools/i8_try.py big > $O/prof_i8.log 2>&1
cat $O/prof_i8.log | tail -3
f=$(ls -t $O/prof_i8/*/*_kernel_stats.csv | head -1)
head -8 $f | cut -c1-150
