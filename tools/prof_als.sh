#!/bin/bash
# Collect the rocprofv3 evidence for the ALS bench on the GPU box (run through gpurun):
#   tools/prof_als.sh <round-tag>
# kernel-trace/stats and each PMC group run as SEPARATE passes (gpurun refuses pmc
# combined with other trace domains; FETCH_SIZE/WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r1}
# PROF_CMD: the command to profile (default: the cfg2 ALS bench); PROF_PASSES: which passes
# ("all" or "short" = stats + the two SQ groups)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD=${PROF_CMD:-"python bench.py --steps 5 --warmup 1 --no-cpu --no-knn --no-topk --no-fit --no-k128 --no-cfg5 --no-cg --no-order-ab"}
PASSES=${PROF_PASSES:-all}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o als -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o als -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc2 -o als -- $CMD > $OUT/pmc2.log 2>&1
if [ "$PASSES" = "all" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o als -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o als -- $CMD > $OUT/pmc4.log 2>&1
fi
find $OUT -name "*.csv" | head -30
du -sh $OUT
