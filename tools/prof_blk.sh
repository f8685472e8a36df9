#!/bin/bash
# the k = 128 and cfg5 parts of tools/prof_round.sh (after a change to csrc/als_blk.hip only)
set -u
R=${1:-r04}
BASE="--no-cpu --no-knn --no-topk --no-fit --no-k128 --no-cfg5 --no-cg"
PROF_CMD="python bench.py --k 128 --steps 5 --warmup 1 $BASE" PROF_PASSES=all bash tools/prof_als.sh ${R}_k128 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_${R}_k128 gpurun_out/${R}_k128
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${R}_cfg5
mkdir -p $OUT
CMD="python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu --no-topk"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg5 -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o cfg5 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o cfg5 -- $CMD > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/${R}_cfg5
ls gpurun_out/${R}_* | head
rm -rf gpurun_out/prof_*
