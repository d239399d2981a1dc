#!/bin/bash
# round 5, final: counters for THIS kernel source (-> profiles/pmc_bench_latest.json), then the bench lines and the configs table
O=gpurun_out/r05z; mkdir -p $O
bash tools/profile_round.sh r05 2>&1 | tail -2
cp gpurun_out/r05_pmc_trace.json profiles/pmc_bench_latest.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver cmd) rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
timeout 900 python bench.py --workload shadow16m --no-cpu > $O/bench_shadow.json 2> $O/bench_shadow.err; echo "bench(shadow16m) rc=$?"
( time timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_g2.json 2> $O/bench_g2.err ) 2>&1 | grep real
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err; echo "configs rc=$?"
python - <<'PY'
import json
for f in ('bench_driver','bench_default'):
    d=json.load(open('gpurun_out/r05z/%s.json'%f))
    r=d['roofline']
    print(f,'value',d['value'],'pipelined',d.get('pipelined',{}).get('value'),'build',d['build']['gpu_build_ms'],'high',d['build']['high_quality']['gpu_build_ms'],'low',d['build']['low_quality']['gpu_build_ms'],'broof',d['build']['roofline']['frac'],'roof',r['frac'],r.get('hbm_counter_frac'), (r.get('valu_from_profile') or {}).get('frac'), 'e2e', d.get('end_to_end',{}).get('value'), d.get('end_to_end',{}).get('frac_of_link_floor'), 'lat', d.get('per_call_latency',{}).get('rtcIntersect1_us_median'), d.get('parity_vs_reference'), 'sustained', d.get('sustained',{}).get('value'))
for f in ('bench_shadow','bench_g2'):
    try:
        d=json.load(open('gpurun_out/r05z/%s.json'%f)); print(f,'value',d['value'],d.get('n_gpus'),d.get('rccl_ranks'),'strong',(d.get('strong') or {}).get('value'))
    except Exception as e: print(f,'parse failed',e)
PY
