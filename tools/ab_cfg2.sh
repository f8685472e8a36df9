#!/bin/bash
# usage: tools/ab_cfg2.sh VAR "v1 v2 ...": the cfg2 headline (k = 64) per value of VAR
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --steps 30 --warmup 3 --no-cpu --no-topk --no-knn --no-fit --no-cg --no-k128 --no-cfg5 "$@" 2>/dev/null | grep '^{' | head -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$var=$v', d.get('value'), d.get('ms_per_step'), d['roofline'].get('avg_launch_ms'))"
done
