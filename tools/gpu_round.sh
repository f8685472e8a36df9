mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_topk.py tests/test_gpu_pipeline.py -m gpu -q -x > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -3 gpurun_out/gputest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_topk_r02e/stats -o topk -- python tools/topk_only.py 64 100 > gpurun_out/prof_topk_r02e.log 2>&1
tail -1 gpurun_out/prof_topk_r02e.log
