for so in "" origin origin+octant octant+origin; do
timeout 300 python tests/gpu_perf.py --reps 6 --tag "sort=$so" ${so:+--sort $so} 2>&1 | grep -A1 PERF | cut -c1-30,95-330
done
