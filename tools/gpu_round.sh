mkdir -p gpurun_out
timeout 600 python tools/knn_variants.py tools/_variants/lkamd_pack16.so > gpurun_out/knn_variants.log 2>&1
cat gpurun_out/knn_variants.log | grep -v amdgpu.ids
