for rep in 1 2; do
for b in 20 16 12 8; do
  MI355_TRACE_BLOCKS_PER_CU=$b timeout 300 python bench.py --no-cpu --steps 60 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('blocks/CU=$b bench', d['value'], 'serial', r['serial']['mrays_per_s'])"
done; done
