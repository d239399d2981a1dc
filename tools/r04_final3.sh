#!/bin/bash
# after the last edit of trace.hip's HOST side (the kernels are the same, the source hash is not): counters again, then the bench lines and the configs table
O=gpurun_out/r04z; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -k "coherent" 2>&1 | tail -1
bash tools/profile_round.sh r04 2>&1 | tail -2
cp gpurun_out/r04_pmc_trace.json profiles/pmc_bench_latest.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver cmd) rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err
python - <<'PY'
import json
for f in ('bench_driver','bench_default'):
    d=json.load(open('gpurun_out/r04z/%s.json'%f))
    print(f,'value',d['value'],'pipelined',d.get('pipelined',{}).get('value'),'build',d['build']['gpu_build_ms'],'high',d['build']['high_quality']['gpu_build_ms'],'roof',d['roofline']['frac'],d['roofline'].get('hbm_counter_from_profile',{}).get('frac'), d['roofline'].get('valu_from_profile',{}).get('frac'), d['roofline']['address_rate']['frac'], 'e2e', d.get('end_to_end',{}).get('value'), d.get('parity_vs_reference'))
PY
grep -E "COHERENT" $O/configs.md | cut -c1-200
