# One gpurun call's worth of work (rewritten per call during development).
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh'
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_als_rhs_order.py tests/test_gpu_iknn_recommend.py tests/test_gpu_als_wb.py -m gpu -q -s -x > gpurun_out/gputest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_new.log
tail -n 5 gpurun_out/gputest_new.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 2 gpurun_out/bench.err
tail -c 6200 gpurun_out/bench.log
