mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --k 128 --steps 10 --no-knn --no-topk > gpurun_out/bench_k128.log 2> gpurun_out/bench_k128.err; echo "rc=$?" >> gpurun_out/bench_k128.err
timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 > gpurun_out/bench_cfg5.log 2> gpurun_out/bench_cfg5.err; echo "rc=$?" >> gpurun_out/bench_cfg5.err
cut -c1-900 gpurun_out/bench_k128.log; tail -2 gpurun_out/bench_k128.err; cut -c1-1500 gpurun_out/bench_cfg5.log; tail -2 gpurun_out/bench_cfg5.err
