mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -3 gpurun_out/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cut -c1-400 gpurun_out/bench.log; tail -2 gpurun_out/bench.err
