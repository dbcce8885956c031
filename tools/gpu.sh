#!/bin/bash
# tools/gpu.sh TIMEOUT 'command' : rebuild the in-tree library first (a stale .so must never travel), then gpurun.
set -e
cd "$(dirname "$0")/.."
# which commit the snapshot is (the GPU box has no .git): read by tools/fallback_suite.sh / profile_round.sh for their logs
{ git rev-parse HEAD; git status --porcelain -- ln-asr_amd include tests bench.py oracle | wc -l; } > .build_commit 2>/dev/null
python ln-asr_amd/build.py > /tmp/lnasr_build.log 2>&1 || { tail -30 /tmp/lnasr_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
