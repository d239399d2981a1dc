for v in "" embree_amd/lib/variant_q10.so embree_amd/lib/variant_w6.so; do
  echo "== lib=$v"
  env ${v:+MI355_LIB=$v} timeout 300 python tests/gpu_perf.py --reps 8 --tag x 2>&1 | grep PERF | cut -c95-260
  env ${v:+MI355_LIB=$v} timeout 300 python bench.py --no-cpu --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench', d['value'], 'conc', r['concurrency'], 'serial', r['serial']['mrays_per_s'])"
done
