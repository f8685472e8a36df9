#!/bin/bash
# rocprofv3 kernel stats of the item-kNN recommend leg: tools/prof_knnrec.sh <tag>
set -u
TAG=${1:-r04_knnrec}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python tools/knn_recommend_only.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o knnrec -- $CMD > $OUT/stats.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/${TAG}
tail -n 1 $OUT/stats.log | head -c 1500
echo
head -n 12 gpurun_out/${TAG}_kernel_stats.csv
