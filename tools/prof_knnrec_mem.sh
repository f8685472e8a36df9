#!/bin/bash
# Memory-path / issue counters of the item-kNN recommend kernels (round 6: what bounds the
# accumulating kernel?): tools/prof_knnrec_mem.sh <tag>  -> gpurun_out/<tag>_mem_counters.csv
set -u
TAG=${1:-r06_knnrec}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_mem
mkdir -p $OUT
CMD="python tools/knn_recommend_only.py"
# (few counters of one block per pass: a request beyond the block's counters aborts rocprofv3 -- and
# the aborted run then hangs until something kills it, hence the timeouts)
run() { d=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -o k -- $CMD > $OUT/$d.log 2>&1; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run c TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
run d TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum
run e SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
run f SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(list)
for p in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    ps = p.split("/")[-2]
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "rec::" not in k and "row_topn" not in k:
            continue
        rows[(ps, k.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("gpurun_out/${TAG}_mem_counters.csv", "w") as f:
    f.write("pass,kernel,counter,mean,count\n")
    for (ps, k, c), v in sorted(rows.items()):
        f.write(f"{ps},{k},{c},{sum(v)/len(v):.6g},{len(v)}\n")
print(open("gpurun_out/${TAG}_mem_counters.csv").read())
PY
