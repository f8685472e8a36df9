#!/bin/bash
# One gpurun call: the rocprofv3 evidence of a round (kernel stats + PMC passes per workload),
# condensed by tools/summarize_prof.py into gpurun_out/<tag>_*  ->  copy to profiles/.
#   tools/prof_round.sh r04
set -u
R=${1:-r06}
BASE="--no-cpu --no-knn --no-topk --no-fit --no-k128 --no-cfg5 --no-cg --no-order-ab"
# cfg2 headline (k = 64): all passes
PROF_PASSES=all bash tools/prof_als.sh ${R}_als_k64 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_${R}_als_k64 gpurun_out/${R}_als_k64
# k = 128 (cfg4's kernel on one GPU): all passes, so that roofline.traffic is not null
PROF_CMD="python bench.py --k 128 --steps 5 --warmup 1 $BASE" PROF_PASSES=all bash tools/prof_als.sh ${R}_k128 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_${R}_k128 gpurun_out/${R}_k128
# cfg5 (k = 256): stats + the FETCH / WRITE passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${R}_cfg5
mkdir -p $OUT
CMD="python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu --no-topk --no-order-ab"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg5 -- $CMD > $OUT/stats.log 2>&1
# (round 5: the SQ groups for cfg5 as well -- VERDICT r4 #4)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o cfg5 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc2 -o cfg5 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o cfg5 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o cfg5 -- $CMD > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/${R}_cfg5
# dense top-K (cfg2) and the item-kNN build / recommend
bash tools/prof_topk.sh ${R} > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_topk_${R} gpurun_out/${R}_topk
bash tools/prof_knn.sh ${R} > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_knn_${R} gpurun_out/${R}_knn
bash tools/prof_knnrec.sh ${R}_knnrec > /dev/null 2>&1
# the als-implicit.toml batch recommend call (round 6)
bash tools/prof_recommend.sh ${R} > /dev/null 2>&1
ls gpurun_out/${R}_* | head -40
# the raw captures are large: keep only the summaries in what gpurun merges back
rm -rf gpurun_out/prof_*
