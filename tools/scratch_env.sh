#!/bin/bash
# env sweep on one library variant: tools/scratch_env.sh <variant> "VAR=val ..." "VAR=val ..." ...
V=$1; shift
for e in "$@"; do
  env $e MI355_LIB=$PWD/embree_amd/lib/variant_$V.so python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline']['per_ray']; print('$V [$e]', 'in_flight', d['value'], 'lone', d['serial']['value'], 'iters', p['wave_iterations'], 'util', p['node_step_simd_util'])"
done
