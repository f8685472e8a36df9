#!/bin/bash
# usage: tools/ab_k128.sh VAR "v1 v2 ...": ML-25M-shaped k = 128 epoch time per value of VAR
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --k 128 --steps 10 --no-cpu --no-topk --no-knn --no-fit --no-cg --no-k128 --no-cfg5 "$@" 2>/dev/null | grep '^{' | head -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$var=$v', d.get('ms_per_step'), d['roofline'].get('frac'), d['roofline'].get('kernel_ms_per_epoch'))"
done
