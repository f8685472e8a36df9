cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_knn_sym
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_knn_sym/stats -o knn -- python tools/knn_only.py > gpurun_out/prof_knn_sym/stats.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_knn_sym/stats/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
