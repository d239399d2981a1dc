#!/bin/bash
# sweep of builder parameters through the device config string: tools/scratch_cfg.sh "cfg1" "cfg2" ...
for c in "$@"; do
  python bench.py --steps 20 --warmup 5 --no-cpu --config "$c" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline']['per_ray']; print('[$c]', 'lone', d['value'], 'pipelined', d['pipelined']['value'], 'nodes', p['nodes'], 'tris', p['triangles'], 'accesses', round(5*p['nodes']+3*p['triangles'],1), 'build', d['build']['gpu_build_ms'], 'bvh MB', d['build']['bvh_bytes']>>20)"
done
