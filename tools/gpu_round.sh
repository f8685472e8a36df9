mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_als.py tests/test_gpu_als_wb.py tests/test_gpu_synth.py tests/test_gpu_als_explicit.py -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
timeout 600 python bench.py --config cfg5 > gpurun_out/bench_cfg5.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_cfg5
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg5 -- python bench.py --config cfg5 --no-cpu --no-topk --steps 2 > $OUT/stats.log 2>&1
python tools/summarize_prof.py $OUT gpurun_out/r03_cfg5 > /dev/null 2>&1
rm -rf $OUT
