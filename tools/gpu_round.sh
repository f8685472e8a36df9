mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -n 3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 1 gpurun_out/bench.err
bash tools/prof_knn.sh r03 > gpurun_out/prof_knn.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_knn_r03 gpurun_out/r03_knn > /dev/null 2>&1
rm -rf gpurun_out/prof_knn_r03
