mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_als.py tests/test_gpu_als_explicit.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py -m gpu -q -x > gpurun_out/gputest_als.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_als.log
tail -4 gpurun_out/gputest_als.log
timeout 600 python tools/als_variants.py tools/_variants/lkamd_nopipe.so tools/_variants/lkamd_nodma.so > gpurun_out/variants_g.log 2>&1
cut -c1-420 gpurun_out/variants_g.log
