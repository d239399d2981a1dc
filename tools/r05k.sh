#!/bin/bash
for w in 1 0; do
  echo "== WHICH=$w robust"; WHICH=$w ROBUST=1 timeout 60 python tests/gpu_devfilter.py 2>&1 | grep -E "calls|rror|fault" | head -3
done
echo "== fast"; timeout 100 python tests/gpu_devfilter.py 2>&1 | grep -E "calls|DEVF|rror|fault" | head -4
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "device_filter" 2>&1 | tail -3
