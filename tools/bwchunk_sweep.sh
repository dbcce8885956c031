#!/bin/bash
# Developer helper (GPU box): config-5 statistics pass against LNASR_BW_CHUNKS (pipelined pass, estimate.hip).
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in 1 2 4 8; do
  echo "== LNASR_BW_CHUNKS=$c"
  LNASR_BW_CHUNKS=$c python3 $R/tools/bench_rows.py bw 2>&1 | grep -v "^{" | grep "Baum\|emit\|mix_mfma\|onchip"
done
