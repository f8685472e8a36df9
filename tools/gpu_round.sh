mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_topk.py tests/test_gpu_scale.py::test_topk_cfg2_all_users_with_exclusions tests/test_gpu_pipeline.py -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -n 6 gpurun_out/gputest.log
timeout 300 python bench.py --no-knn --no-fit --no-k128 --no-cfg5 --steps 5 > gpurun_out/bench_taumask.log 2>&1
LK_TOPK_TAU_MASK=0 timeout 300 python bench.py --no-knn --no-fit --no-k128 --no-cfg5 --no-cpu --steps 5 > gpurun_out/bench_notaumask.log 2>&1
python - <<'PY'
import json
for f in ('gpurun_out/bench_taumask.log','gpurun_out/bench_notaumask.log'):
    l=[x for x in open(f) if x.startswith('{')][-1]; d=json.loads(l); t=d['topk']
    print(f, t['value'], t['roofline']['frac'], t.get('parity'))
PY
