#!/bin/bash
# rocprofv3 evidence for the als-implicit.toml batch recommend call (run through gpurun):
#   tools/prof_recommend.sh <tag>     -> gpurun_out/<tag>_recommend_{kernel_stats,dispatches,counters}.csv
# kernel-trace/stats and each PMC group are SEPARATE passes (gpurun refuses pmc + other domains).
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_recommend_$TAG
mkdir -p $OUT
CMD=${PROF_CMD:-"python tools/recommend_profile.py 10000 short"}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o rec -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o rec -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -o rec -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -o rec -- $CMD > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/${TAG}_recommend
tail -3 $OUT/stats.log
rm -rf $OUT
