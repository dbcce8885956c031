#!/bin/bash
# Runs on the GPU box: the whole GPU suite once per group of fallback switches, so that every layer
# under the tuned kernels is itself shown parity-green.   gpurun -- 'bash tools/fallback_suite.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "# commit $(head -1 .build_commit 2>/dev/null || echo unknown), $(sed -n 2p .build_commit 2>/dev/null || echo '?') uncommitted source file(s), $(date -u +%Y-%m-%dT%H:%MZ)"
rc=0
run() {
    echo "== $*"
    env "$@" timeout -k 10 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
    [ ${PIPESTATUS[0]} -eq 0 ] || rc=1
}
run LNASR_NO_PAD=1 LNASR_NO_I8EMIT=1
run LNASR_NO_MFMA=1 LNASR_NO_MFMA_STATS=1
run LNASR_NO_TINYVIT=1 LNASR_NO_TINYFWD=1 LNASR_NO_COLFWD=1 LNASR_NO_REGLU=1 LNASR_NO_TILEPACK=1
run LNASR_NO_TINYVIT=1 LNASR_NO_TINYFWD=1 LNASR_NO_REGVIT=1 LNASR_NO_REGFWD=1 LNASR_NO_T256=1 LNASR_NO_LDSBT=1 LNASR_VIT=a
run LNASR_NO_STREAM=1 LNASR_NO_FRAMEPAR=1 LNASR_NO_T256=1 LNASR_F32_GENERIC=1 LNASR_NO_FFT512=1 LNASR_NO_PIPELINE=1
run LNASR_NO_PINNED_RING=1 LNASR_H2D_THREADS=2 LNASR_FWD_ONE_MAXU=0 LNASR_NO_LATFWD=1 LNASR_CHASE_MAXU=0
run LNASR_I8_MIN_FRAMES=0               # the integer matrix-core bank for every model it takes, whatever the batch size
run LNASR_CHECK_PTRS=1                 # every pointer a launch helper is handed is verified to be device memory
exit $rc
