#!/bin/bash
# rocprofv3 kernel stats of the item-kNN recommend leg: tools/prof_knnrec.sh <tag>
set -u
TAG=${1:-r04_knnrec}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python tools/knn_recommend_only.py"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o knnrec -- $CMD > $OUT/stats.log 2>&1
# (round 5: counter passes as well -- VERDICT r4 #6)
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o knnrec -- $CMD > $OUT/pmc1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc2 -o knnrec -- $CMD > $OUT/pmc2.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o knnrec -- $CMD > $OUT/pmc3.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o knnrec -- $CMD > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/${TAG}
tail -n 1 $OUT/stats.log | head -c 1500
echo
head -n 12 gpurun_out/${TAG}_kernel_stats.csv
