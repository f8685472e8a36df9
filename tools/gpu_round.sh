mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_iknn_score.py tests/test_gpu_uknn.py tests/test_gpu_seam.py tests/test_gpu_pipeline.py -m gpu -q -s > gpurun_out/gputest_knn.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_knn.log
tail -n 3 gpurun_out/gputest_knn.log
LK_KNN_SCORE_LISTS=0 timeout 600 python bench.py --steps 3 --no-topk --no-fit --no-k128 --no-cfg5 --no-cg > gpurun_out/bench_knn_slot.log 2> gpurun_out/bench_knn_slot.err; echo "bench rc=$?"
