#!/bin/bash
# tools/gpu.sh TIMEOUT 'command' : rebuild the in-tree library first (a stale .so must never travel), then gpurun.
set -e
cd "$(dirname "$0")/.."
python ln-asr_amd/build.py > /tmp/lnasr_build.log 2>&1 || { tail -30 /tmp/lnasr_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
