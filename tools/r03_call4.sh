#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "crown_small or far_from_the_origin or bit_identical or spatial_split or several_gpus or detach or rejected_candidate or more_gpus" -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | head -40
timeout 900 python bench.py --inprocess-gpus 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d/bench.json'))
print('value',d['value'],'serial',d.get('serial',{}).get('value'),'build',d['build']['gpu_build_ms'],d['build']['mprims_per_s_gpu'])
print('roof', {k:d['roofline'][k] for k in ('achieved','frac','peak_measured','launches_in_flight')})
print('per_ray', d['roofline']['per_ray']['nodes'], d['roofline']['per_ray']['triangles'])
print('ref visits', d.get('reference_visits'))
print('latency', d.get('per_call_latency'))
print('multi', d.get('in_process_multi_gpu'))
print('e2e', d.get('end_to_end'))
print('parity', d.get('parity_vs_reference'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
