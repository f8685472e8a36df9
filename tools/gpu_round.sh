# One gpurun call's worth of work (rewritten per call during development): the full GPU suite, the
# default bench line and the smoke entry point.  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh'
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 1 gpurun_out/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -n 2 gpurun_out/smoke.log
