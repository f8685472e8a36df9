# one gpurun call: GPU tests, smoke, bench line, A/B runs (results under gpurun_out/)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -3 gpurun_out/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_main.log 2>&1; echo "smoke(main) rc=$?" >> gpurun_out/smoke_main.log
tail -2 gpurun_out/smoke.log gpurun_out/smoke_main.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -1 gpurun_out/bench.err
LK_TOPK_FUSED_ROWS=65536 timeout 300 python bench.py --no-knn --no-fit --no-cpu --no-k128 --no-cfg5 --steps 5 > gpurun_out/bench_topk_rows65536.log 2>&1
timeout 300 python tools/blk_variants.py 128 tools/_variants/lkamd_blk_w8_3.so tools/_variants/lkamd_blk_w8_2.so > gpurun_out/blk128.log 2>&1
timeout 300 python tools/blk_variants.py 256 tools/_variants/lkamd_blk_w1.so > gpurun_out/blk256.log 2>&1
tail -3 gpurun_out/blk128.log gpurun_out/blk256.log
du -sh gpurun_out
