mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
PROF_CMD="python tools/cg_only.py 64 1e-6" bash tools/prof_als.sh r03_cg > gpurun_out/prof_cg.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r03_cg gpurun_out/r03_cg_k64 > /dev/null 2>&1
rm -rf gpurun_out/prof_r03_cg
cp gpurun_out/r03_cg_k64_*.csv profiles/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 1 gpurun_out/bench.err
