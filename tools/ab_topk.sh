#!/bin/bash
# usage: tools/ab_topk.sh VAR "v1 v2 ...": the cfg2 dense top-100 leg per value of VAR
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --steps 3 --warmup 1 --no-cpu --no-knn --no-fit --no-k128 --no-cfg5 --no-cg "$@" 2>/dev/null | grep '^{' | head -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d['topk']; print('$var=$v', t.get('value'), t['roofline'].get('frac'))"
done
