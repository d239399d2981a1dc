#!/bin/bash
# round 5, GPU call 1: instruction costs of the candidates, then the trace-kernel variants one process each (same box), then the GPU tests on the product library
O=gpurun_out/r05a; mkdir -p $O
timeout 120 tools/valu_bench 40 > $O/valu_new.txt 2>&1
timeout 60 tools/valu_bench 12 2>&1 | head -2 >> $O/valu_new.txt
for v in base sel push async; do
  MI355_LIB=embree_amd/lib/variant_$v.so timeout 200 python tests/gpu_knobs.py $v 2>&1 | grep -E "KNOBS|rror" >> $O/knobs.log
done
timeout 200 python tests/gpu_knobs.py all 2>&1 | grep -E "KNOBS|rror" >> $O/knobs.log
MI355_REFILL_MIN=8 timeout 200 python tests/gpu_knobs.py all_g8 2>&1 | grep -E "KNOBS|rror" >> $O/knobs.log
MI355_PUSH_ROUNDS=8 timeout 200 python tests/gpu_knobs.py all_p8 2>&1 | grep -E "KNOBS|rror" >> $O/knobs.log
MI355_LIB=embree_amd/lib/variant_base.so timeout 200 python tests/gpu_knobs.py base_again 2>&1 | grep -E "KNOBS|rror" >> $O/knobs.log
cat $O/knobs.log
timeout 200 python tests/gpu_perf.py --tag all 2>&1 | tail -3 > $O/perf_all.log; cat $O/perf_all.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
grep -E "cndmask|cmp_e|sdwa|bitop|dpp|denormal" $O/valu_new.txt | grep "SIMD 4"
