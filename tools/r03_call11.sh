#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "bench --gpus 2 rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r03n/bench_gpus2.json'))
    print({k:d[k] for k in ('value','n_gpus','ranks','distinct_gpus','rccl_ranks','ms_per_step','scaling')}, d.get('gather'))
except Exception as e:
    print('no line', e); print(open('gpurun_out/r03n/bench_gpus2.err').read()[-2000:])
PY
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -q --tb=short -k "shadow16m" -rf -s > $O/pytest_shadow.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_shadow.log
grep -E "^FAILED|^ERROR|passed|failed|^E  |whole job" $O/pytest_shadow.log | head
