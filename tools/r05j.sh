#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
timeout 100 python tests/gpu_devfilter.py 2>&1 | grep -E "calls|DEVF|rror" | head -4
ROBUST=1 timeout 100 python tests/gpu_devfilter.py 2>&1 | grep -E "calls|rror|fault" | head -4
timeout 200 python -m pytest tests/test_gpu_round4.py -m gpu -q -k "latency" -s 2>&1 | grep -E "rtcIntersect1|passed|failed"
timeout 100 python tests/gpu_latency.py 2>&1 | tail -4
