# bench + smoke only (the full round script is tools/gpu_round.sh)
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 2 gpurun_out/bench.err
tail -n 1 gpurun_out/bench.log | head -c 6500
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -n 3 gpurun_out/smoke.log
