mkdir -p gpurun_out
tools/ub/permlane_swap
timeout 600 python -m pytest tests/test_gpu_als.py tests/test_gpu_als_explicit.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py -m gpu -q -x > gpurun_out/gputest_als.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_als.log
tail -4 gpurun_out/gputest_als.log
timeout 600 python tools/als_variants.py tools/_variants/lkamd_old3.so tools/_variants/lkamd_h4r2.so tools/_variants/lkamd_h3r4.so tools/_variants/lkamd_phases.so tools/_variants/lkamd_h3phases.so > gpurun_out/variants_d.log 2>&1
cat gpurun_out/variants_d.log
