timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tests/gpu_perf.py --reps 8 --tag pkfma 2>&1 | grep -A1 PERF | cut -c1-330
timeout 300 python tests/gpu_overlap.py 2>&1 | grep -E "streams=(1|4)"
