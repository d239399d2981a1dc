#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
timeout 200 tests/golden/_bin/ref_verify --no-colors --sequential --flatten --intensity 0.2 --run ".*\.instancing\.instancing\..*" > $O/inst.txt 2>&1; echo "rc=$?"
grep -c PASSED $O/inst.txt; grep FAILED $O/inst.txt | head -8 | cut -c1-200; grep -E "rror" $O/inst.txt | sort | uniq -c | head
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -k "device_filter_function" 2>&1 | tail -30 > $O/pytest_df.log; cat $O/pytest_df.log | cut -c1-250
