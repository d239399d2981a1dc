#!/bin/bash
# round 5, GPU call 3: one 16-byte load for the third word of a triangle record (KEEP4), and what goes with it
O=gpurun_out/r05c; mkdir -p $O
k() { tag=$1; shift; env "$@" timeout 90 python tests/gpu_knobs.py $tag 2>&1 | grep -E "KNOBS|rror|fault" >> $O/knobs.log; }
k k4
k sp MI355_LIB=embree_amd/lib/variant_sp.so
k k4_p8 MI355_PUSH_ROUNDS=8
k k4_noslp MI355_LIB=embree_amd/lib/variant_noslp.so
k k4_noslp_p8 MI355_LIB=embree_amd/lib/variant_noslp.so MI355_PUSH_ROUNDS=8
k k4_w5np MI355_LIB=embree_amd/lib/variant_w5np.so
k k4_q512 MI355_LIB=embree_amd/lib/variant_q512.so
k k4_q512_p8 MI355_LIB=embree_amd/lib/variant_q512.so MI355_PUSH_ROUNDS=8
k k4_g32 MI355_REFILL_MIN=32
k k4_b12 MI355_TRACE_BLOCKS_PER_CU=12
cat $O/knobs.log
MI355_PUSH_ROUNDS=8 timeout 100 python tests/gpu_perf.py --tag k4_p8 2>&1 | tail -2 > $O/perf_k4.log; cat $O/perf_k4.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 > $O/pytest.log; cat $O/pytest.log
