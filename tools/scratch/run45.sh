for rep in 1 2 3; do
for v in "" embree_amd/lib/variant_b64.so; do
  env ${v:+MI355_LIB=/root/repo/$v} timeout 300 python bench.py --no-cpu --steps 60 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lib=$v bench', d['value'], 'serial', r['serial']['mrays_per_s'])"
done; done
