#!/bin/bash
# rocprofv3 evidence for the item-kNN build (run through gpurun): tools/prof_knn.sh <tag>
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_knn_$TAG
mkdir -p $OUT
CMD="python tools/knn_only.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o knn -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o knn -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc2 -o knn -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o knn -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o knn -- $CMD > $OUT/pmc4.log 2>&1
du -sh $OUT
