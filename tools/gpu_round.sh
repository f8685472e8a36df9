# one gpurun call: GPU tests, bench line, top-K profile (results under gpurun_out/)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -3 gpurun_out/gputest.log
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -1 gpurun_out/bench.err
timeout 400 bash tools/prof_topk.sh r03 > gpurun_out/prof_topk.log 2>&1
