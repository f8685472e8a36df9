mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 3 gpurun_out/gputest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_r03b_knnscore
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03b_knnscore/stats -o als -- python bench.py --steps 3 --no-topk --no-fit --no-k128 --no-cfg5 --no-cg --no-cpu > gpurun_out/prof_r03b_knnscore/stats.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r03b_knnscore gpurun_out/r03b_knnscore > /dev/null 2>&1
rm -rf gpurun_out/prof_r03b_knnscore
cp gpurun_out/r03b_knnscore_*.csv profiles/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -n 1 gpurun_out/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -n 2 gpurun_out/smoke.log
