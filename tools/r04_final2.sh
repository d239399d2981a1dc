#!/bin/bash
# end-of-round measurements after the last builder changes (the traversal kernel is what tools/r04_final.sh profiled: same source hash)
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver cmd) rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/r04_tl_raw -- python $OLDPWD/tests/gpu_build_only.py "" 4 > /dev/null 2>&1 )
python tools/ktimeline.py gpurun_out/r04_tl_raw v > $O/timeline_medium.txt 2>&1; rm -rf gpurun_out/r04_tl_raw
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/r04_tl_raw -- python $OLDPWD/tests/gpu_build_only.py "" 3 2 > /dev/null 2>&1 )
python tools/ktimeline.py gpurun_out/r04_tl_raw > $O/timeline_high.txt 2>&1; rm -rf gpurun_out/r04_tl_raw
PP=1 TREEHASH=1 timeout 200 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -E "TREEHASH|BUILD"
TREEHASH=1 timeout 200 python tests/gpu_build_only.py "" 6 2>&1 | grep -E "TREEHASH|BUILD"
tail -24 $O/timeline_high.txt | head -8
python - <<'PY'
import json
for f in ('bench_driver','bench_default'):
    d=json.load(open('gpurun_out/r04z/%s.json'%f))
    print(f,'value',d['value'],'pipelined',d.get('pipelined',{}).get('value'),'build',d['build']['gpu_build_ms'],d['build']['mprims_per_s_gpu'],'high',d['build']['high_quality'],'roof',d['roofline']['frac'],d['roofline'].get('hbm_counter_from_profile',{}).get('frac'), d['roofline'].get('valu_from_profile',{}).get('frac'), d['roofline']['address_rate']['frac'], 'e2e', d.get('end_to_end',{}).get('value'), 'lat', d.get('per_call_latency',{}).get('rtcIntersect1_us_median'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('parity_vs_reference'))
PY
