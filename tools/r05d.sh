#!/bin/bash
# round 5, GPU call 4: waits apart (node loads unconditional + stay in flight over the triangle block, stack pop as ds_read), then the whole GPU suite and the bench
O=gpurun_out/r05d; mkdir -p $O
k() { tag=$1; shift; env "$@" timeout 90 python tests/gpu_knobs.py $tag 2>&1 | grep -E "KNOBS|rror|fault" >> $O/knobs.log; }
k cur
k prev MI355_LIB=embree_amd/lib/variant_prev.so
k cur_again
k cur_p5 MI355_PUSH_ROUNDS=5
k cur_p12 MI355_PUSH_ROUNDS=12
k cur_g8 MI355_REFILL_MIN=8
cat $O/knobs.log
timeout 100 python tests/gpu_perf.py --tag cur 2>&1 | tail -2 > $O/perf_cur.log; cat $O/perf_cur.log
timeout 100 python tests/gpu_perf.py --tag cur_any --any 2>&1 | tail -2 >> $O/perf_cur.log
timeout 100 python tests/gpu_perf.py --tag cur_pp --powerplant 2>&1 | tail -2 >> $O/perf_cur.log
MI355_LIB=embree_amd/lib/variant_prev.so timeout 100 python tests/gpu_perf.py --tag prev_any --any 2>&1 | tail -2 >> $O/perf_cur.log
MI355_LIB=embree_amd/lib/variant_prev.so timeout 100 python tests/gpu_perf.py --tag prev_pp --powerplant 2>&1 | tail -2 >> $O/perf_cur.log
MI355_LIB=embree_amd/lib/variant_base.so timeout 100 python tests/gpu_perf.py --tag base_any --any 2>&1 | tail -2 >> $O/perf_cur.log
MI355_LIB=embree_amd/lib/variant_base.so timeout 100 python tests/gpu_perf.py --tag base_pp --powerplant 2>&1 | tail -2 >> $O/perf_cur.log
grep PERF $O/perf_cur.log | cut -c1-200
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d/bench_driver.json'))
print('value',d['value'],'pipelined',d.get('pipelined',{}).get('value'),'build',d['build']['gpu_build_ms'],d.get('parity_vs_reference'))
PY
