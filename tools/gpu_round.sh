mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_cfg5
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg5 -- python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu --no-topk > $OUT/stats.log 2>&1
tail -2 $OUT/stats.log | cut -c1-300
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_cfg5/stats/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
