# One gpurun call's worth of work (rewritten per call during development).
# Usage: gpurun --timeout 2400 -- 'bash tools/gpu_round.sh'
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
grep -E "passed|failed|error" gpurun_out/gputest.log | tail -n 5
grep -E "^FAILED|^k = |at-scale" gpurun_out/gputest.log | head -20
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 2 gpurun_out/bench.err
tail -n 1 gpurun_out/bench.log | head -c 6500
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -n 3 gpurun_out/smoke.log
