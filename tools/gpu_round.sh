mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_topk.py tests/test_gpu_scale.py tests/test_gpu_pipeline.py tests/test_gpu_seam.py -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
timeout 300 python bench.py --no-knn --no-fit --no-k128 --no-cfg5 --steps 5 > gpurun_out/bench_fused.log 2>&1
LK_TOPK_FUSE_SELECT=0 timeout 300 python bench.py --no-knn --no-fit --no-k128 --no-cfg5 --no-cpu --steps 5 > gpurun_out/bench_unfused.log 2>&1
timeout 400 python tools/blk_variants.py 128 tools/_variants/lkamd_r0.so tools/_variants/lkamd_r2_1.so tools/_variants/lkamd_r4w3_4.so tools/_variants/lkamd_r8w3.so > gpurun_out/blk128.log 2>&1
timeout 400 python tools/blk_variants.py 256 tools/_variants/lkamd_r0.so tools/_variants/lkamd_r2_1.so tools/_variants/lkamd_r4w3_4.so > gpurun_out/blk256.log 2>&1
grep "^{" gpurun_out/blk128.log gpurun_out/blk256.log
