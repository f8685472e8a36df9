#!/bin/bash
# rocprofv3 evidence for dense scoring + top-N (run through gpurun): tools/prof_topk.sh <tag>
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_topk_$TAG
mkdir -p $OUT
CMD="python tools/topk_only.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o topk -- $CMD > $OUT/stats.log 2>&1
tail -1 $OUT/stats.log
