mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_pipeline.py tests/test_gpu_seam.py -m gpu -q -x > gpurun_out/gputest_topk.log 2>&1; tail -3 gpurun_out/gputest_topk.log
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err; tail -1 gpurun_out/bench.err
PROF_CMD="python tools/topk_only.py 64 100" timeout 500 bash tools/prof_topk.sh r02c > gpurun_out/prof_topk.log 2>&1
