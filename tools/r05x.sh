#!/bin/bash
O=gpurun_out/r05x; mkdir -p $O
for w in 5 5; do
timeout 300 python bench.py --steps 20 --warmup $w --no-cpu --sustain 0 > $O/bench_w$w.json 2> $O/bench_w$w.err
python - <<PY
import json
d=json.loads(open('$O/bench_w$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('w$w', d['value'], d['pipelined']['value'], r['kernel_ms_avg_overlapping'], r['kernel_ms_min'], r['kernel_ms_first4'], r['kernel_ms_last4'])
PY
done
