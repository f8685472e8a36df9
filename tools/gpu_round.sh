# one gpurun call: GPU tests, smoke, bench line, A/B runs (results under gpurun_out/)
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_main.log 2>&1; echo "smoke(main) rc=$?" >> gpurun_out/smoke_main.log
tail -n 2 gpurun_out/smoke_main.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 1 gpurun_out/bench.err
timeout 300 python tools/download_bench.py > gpurun_out/download.log 2>&1
tail -n 12 gpurun_out/download.log
du -sh gpurun_out
