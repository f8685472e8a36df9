mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_iknn_score.py tests/test_gpu_uknn.py tests/test_gpu_seam.py tests/test_gpu_pipeline.py -m gpu -q -s > gpurun_out/gputest_knn.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_knn.log
tail -n 3 gpurun_out/gputest_knn.log
timeout 600 python -m pytest tests/test_gpu_als.py -m gpu -q -s -k "cg" > gpurun_out/gputest_cg.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_cg.log
tail -n 3 gpurun_out/gputest_cg.log
rm -f gpurun_out/cg_only.log
for k in 64 128 256; do timeout 300 python tools/cg_only.py $k 1e-6 >> gpurun_out/cg_only.log 2>&1; done
cat gpurun_out/cg_only.log
timeout 600 python bench.py --steps 5 --no-topk --no-fit --no-k128 --no-cfg5 --no-cg > gpurun_out/bench_knn.log 2> gpurun_out/bench_knn.err; echo "bench rc=$?"
