cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_acc_ef
mkdir -p $OUT
CMD="python tools/knn_recommend_only.py"
run() { d=$1; shift; LK_REC_OVERLAP=0 timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -o k -- $CMD > $OUT/$d.log 2>&1; }
run e SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
run f SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(list)
for p in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "score_acc" not in k: continue
        rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(rows.items()):
    print(c, f"{sum(v)/len(v):.5g}", len(v))
PY
