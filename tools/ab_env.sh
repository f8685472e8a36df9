#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [bench args]: cfg5 epoch time + kernel split per value of VAR
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --config cfg5 --no-cpu --no-topk "$@" 2>/dev/null | grep '^{' | head -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$var=$v', d.get('ms_per_step'), d['roofline'].get('kernel_ms_per_epoch'))"
done
