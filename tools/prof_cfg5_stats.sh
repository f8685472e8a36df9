#!/bin/bash
# cfg5 kernel stats only (one rocprofv3 pass): gpurun_out/<tag>_cfg5_kernel_stats.csv
R=${1:-tmp}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${R}_cfg5
mkdir -p $OUT
CMD="python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu --no-topk"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg5 -- $CMD > $OUT/stats.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/${R}_cfg5
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${R}_cfg5_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), "%9.3f ms avg" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
PY
rm -rf gpurun_out/prof_*
