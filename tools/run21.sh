for t in 256 512 2048 4096 8192 16384; do
  timeout 300 python tests/gpu_perf.py --reps 2 --tag small$t --config small_threshold=$t 2>&1 | grep PERF | cut -c1-140
done
