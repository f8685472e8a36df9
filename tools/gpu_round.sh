mkdir -p gpurun_out
PROF_CMD="python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu --no-topk" timeout 1500 bash tools/prof_als.sh r02_cfg5b > gpurun_out/prof_cfg5b.log 2>&1
tail -3 gpurun_out/prof_cfg5b.log
