mkdir -p gpurun_out
timeout 800 python tools/knn_variants.py tools/_variants/lkamd_chunk128.so tools/_variants/lkamd_chunk128r4.so > gpurun_out/knn_variants.log 2>&1
cat gpurun_out/knn_variants.log | grep -v amdgpu.ids
