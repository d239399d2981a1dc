#!/bin/bash
# round 5, last GPU call: the whole GPU suite, smoke(), the driver's bench command, the in-process 2-replica leg
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 > $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --sustain 0 --inprocess-gpus 2 > $O/bench_multi.json 2> $O/bench_multi.err; echo "bench multi rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_final/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], 'pipelined', d['pipelined']['value'], 'build', d['build']['gpu_build_ms'], 'parity', d['parity_vs_reference'].get('unexplained'), 'traffic', d['roofline'].get('traffic'), 'first4', d['roofline'].get('kernel_ms_first4'))
m=json.loads(open('gpurun_out/r05_final/bench_multi.json').read().strip().splitlines()[-1])
print('multi', m.get('in_process_multi_gpu'))
PY
