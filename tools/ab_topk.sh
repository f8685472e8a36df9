#!/bin/bash
# one gpurun call: fused top-N tests, the stage-1 / stage-3 A/B, and its kernel trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != 1 ]; then timeout 900 python -m pytest tests/test_gpu_topk.py -x -q 2>&1 | tail -12; fi
timeout 500 python tools/topk_ab.py 64 100 4 2>&1 | tail -16
if [ -f tools/_variants/lkamd_wselph.so ]; then
  LK_AMD_LIBRARY=$PWD/tools/_variants/lkamd_wselph.so timeout 500 python tools/topk_ab.py 64 100 2 2>&1 | grep -v stage1 | tail -12
fi
if [ "${SKIP_TRACE:-0}" = 1 ]; then exit 0; fi
OUT=gpurun_out/topk_ab
rm -rf $OUT; mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace -d $OUT -- python tools/topk_ab.py 64 100 1 > $OUT/log.txt 2>&1
python tools/trace_stats.py $OUT score_ sample_ cmax_ cand_select row_topn 2>&1 | tail -30
find $OUT -name "*.db" -delete
