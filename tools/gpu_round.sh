mkdir -p gpurun_out
rm -f gpurun_out/ks_ab.log
for cfg in "1024 8" "1024 16" "512 8" "512 16" "256 16" "256 32" "2048 8"; do
set -- $cfg
echo "heavy $1 split $2" >> gpurun_out/ks_ab.log
LK_KNN_SCORE_HEAVY=$1 LK_KNN_SCORE_SPLIT=$2 timeout 300 python bench.py --steps 3 --no-topk --no-fit --no-k128 --no-cfg5 --no-cg --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['knn']['batch_score']['seconds'])
" >> gpurun_out/ks_ab.log
done
cat gpurun_out/ks_ab.log
