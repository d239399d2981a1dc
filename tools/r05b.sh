#!/bin/bash
# round 5, GPU call 2: the push scan (fixed), pushRounds / refillMin with it, five waves per SIMD
O=gpurun_out/r05b; mkdir -p $O
k() { tag=$1; shift; env "$@" timeout 90 python tests/gpu_knobs.py $tag 2>&1 | grep -E "KNOBS|rror|fault" >> $O/knobs.log; }
k sp
k base MI355_LIB=embree_amd/lib/variant_base.so
k sp_p8 MI355_PUSH_ROUNDS=8
k sp_p12 MI355_PUSH_ROUNDS=12
k sp_p24 MI355_PUSH_ROUNDS=24
k sp_g8 MI355_REFILL_MIN=8
k sp_g8_p12 MI355_REFILL_MIN=8 MI355_PUSH_ROUNDS=12
k w5np MI355_LIB=embree_amd/lib/variant_w5np.so
k w5np_p12 MI355_LIB=embree_amd/lib/variant_w5np.so MI355_PUSH_ROUNDS=12
k w5 MI355_LIB=embree_amd/lib/variant_w5.so
k w5_p12 MI355_LIB=embree_amd/lib/variant_w5.so MI355_PUSH_ROUNDS=12
cat $O/knobs.log
timeout 100 python tests/gpu_perf.py --tag sp 2>&1 | tail -2 > $O/perf_sp.log; cat $O/perf_sp.log
MI355_PUSH_ROUNDS=12 timeout 100 python tests/gpu_perf.py --tag sp_p12 2>&1 | tail -2 > $O/perf_sp_p12.log; cat $O/perf_sp_p12.log
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
