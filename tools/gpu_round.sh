mkdir -p gpurun_out
timeout 600 python tools/knn_variants.py 4096 2048 tools/_variants/lkamd_ring16.so tools/_variants/lkamd_ring32.so > gpurun_out/knn_variants.log 2>&1
grep "^{" gpurun_out/knn_variants.log
