mkdir -p gpurun_out
for k in 32 128 256; do timeout 300 python tools/topk_variants.py $k 100 tools/_variants/lkamd_slab.so > gpurun_out/topk_slab_$k.log 2>&1; grep "^{" gpurun_out/topk_slab_$k.log; done
