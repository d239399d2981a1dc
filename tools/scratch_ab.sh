#!/bin/bash
# A/B of library variants: tools/scratch_ab.sh name1 name2 ...   (embree_amd/lib/variant_<name>.so)
for v in "$@"; do
  MI355_LIB=$PWD/embree_amd/lib/variant_$v.so python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline']['per_ray']; print('$v', 'in_flight', d['value'], 'lone', d['serial']['value'], 'kernel_ms', d['roofline']['kernel_ms_avg'], 'nodes', p['nodes'], 'tris', p['triangles'], 'recs', p.get('records'), 'acc', d['roofline']['address_rate']['per_ray'], 'addr_frac', d['roofline']['address_rate'].get('frac'), 'util', p['node_step_simd_util'], 'iters', p['wave_iterations'], 'empty', p['empty_node_visits'])"
done
