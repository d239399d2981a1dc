#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver cmd) rc=$?"
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_reference_suite.py -m gpu -q 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05y/bench_driver.json'))
print('value',d['value'],'pipelined',d['pipelined']['value'],'lat',d['per_call_latency'],'gpu_over_cpu',d.get('gpu_over_cpu'),'traffic_is' in d['roofline'])
PY
