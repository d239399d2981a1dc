#!/bin/bash
# end-of-round measurements: the driver's bench command, the default one, kernel stats + PMC of the same kernel source, the all-configs table
O=gpurun_out/r03z; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver cmd) rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
bash tools/profile_round.sh r03 2>&1 | tail -3
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err
python - <<'PY'
import json
for f in ('bench_driver','bench_default'):
    d=json.load(open('gpurun_out/r03z/%s.json'%f))
    print(f,'value',d['value'],'serial',d.get('serial',{}).get('value'),'build',d['build']['gpu_build_ms'],d['build']['mprims_per_s_gpu'],'roof',d['roofline']['frac'],d['roofline'].get('hbm_counter_from_profile',{}).get('frac'), d['roofline'].get('valu_from_profile',{}).get('frac'), d['roofline']['address_rate']['frac'], 'e2e', d.get('end_to_end',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('parity_vs_reference'))
PY
