#!/bin/bash
# rocprofv3 evidence for dense scoring + top-N (run through gpurun): tools/prof_topk.sh <tag>
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_topk_$TAG
mkdir -p $OUT
# default: the bench workload itself (trained factors, the users' own items excluded)
CMD=${PROF_CMD:-"python bench.py --steps 3 --warmup 1 --no-cpu --no-knn --no-fit --no-k128 --no-cfg5 --no-cg --no-order-ab"}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o topk -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o topk -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o topk -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o topk -- $CMD > $OUT/pmc4.log 2>&1
tail -1 $OUT/stats.log
du -sh $OUT
