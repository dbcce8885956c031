#!/bin/bash
# tools/build_ref.sh GIT_REF NAME : export an earlier commit's tree (package, bench.py, oracle, tools) to _ab/NAME/ and build its
# library there, so that an A/B of two revisions runs inside ONE gpurun call (`python _ab/NAME/bench.py ...` against
# `python bench.py ...`).  _ab/ is git-ignored and travels to the GPU box.
set -e
cd "$(dirname "$0")/.."
ref="$1"; name="$2"
rm -rf "_ab/$name"
mkdir -p "_ab/$name"
git archive "$ref" ln-asr_amd include oracle tools bench.py __graft_entry__.py profiles/pmc_traffic.json | tar -x -C "_ab/$name"
python "_ab/$name/ln-asr_amd/build.py" > "_ab/$name/build.log" 2>&1 || { tail -30 "_ab/$name/build.log"; exit 1; }
rm -rf "_ab/$name/ln-asr_amd/build"
echo "_ab/$name"
