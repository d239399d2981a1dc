#!/bin/bash
# round 5, GPU call 5: the reference's own test programs against the library (exploration), new tests, dry run of the 8-rank bench, build A/B
O=gpurun_out/r05e; mkdir -p $O
V=tests/golden/_bin/ref_verify
(timeout 60 $V --no-colors --print-tests > $O/verify_tests.txt 2>&1; echo "print-tests rc=$?")
for pat in "create_device" ".*multiple_devices" ".*types_test" ".*get_bounds.triangles" ".*get_bounds.quads" ".*get_user_data" ".*buffer_stride.triangles" ".*buffer_stride.quads" \
           ".*empty_scene.*" ".*empty_geometry.*" ".*triangle_hit.*" ".*quad_hit.*" ".*inactive_rays.*" ".*ray_masks.*" ".*backfacing.*" ".*small_triangle_hit.*" ".*ray_alignment_test.*" \
           ".*watertight_triangles\..*" ".*watertight_quads\..*" ".*enable_disable_geometry.*" ".*disable_detach_geometry.*" ".*new_delete_geometry.*" ".*build_garbage_geom.*" \
           ".*build\..*" ".*overlapping_primitives.*" ".*user_geometry_id.*" ".*intersection_filter.*|.*filter.*" ".*instancing.*"; do
  s=$(date +%s.%N)
  timeout 150 $V --no-colors --sequential --intensity 0.2 --run "$pat" > $O/v.tmp 2>&1; rc=$?
  e=$(date +%s.%N)
  echo "== $pat rc=$rc $(echo "$e - $s" | bc) s" >> $O/verify_runs.txt
  grep -E "PASSED|FAILED|SKIPPED|rror|terminate|Segmentation|fault" $O/v.tmp | sort | uniq -c | sort -rn | head -12 >> $O/verify_runs.txt
done
timeout 100 $V --no-colors --sequential --intensity 0.2 --run ".*update.*Fast.MediumQuality.Intersect1" > $O/v_update.txt 2>&1; echo "update rc=$?" >> $O/verify_runs.txt; tail -5 $O/v_update.txt >> $O/verify_runs.txt
cat $O/verify_runs.txt | cut -c1-160
(cd $O && timeout 200 ../../tests/golden/_bin/ref_triangle_geometry --compare ../../tests/golden/models/triangle_geometry.exr -o tg.ppm > tg.log 2>&1; echo "triangle_geometry rc=$?"; tail -5 tg.log)
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -m gpu -x -q --durations=8 2>&1 | tail -16 > $O/pytest.log; cat $O/pytest.log
( time timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench_g8.json 2> $O/bench_g8.err ) 2>&1 | grep real; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r05e/bench_g8.json'))
    print('g8 value',d['value'],'n_gpus',d['n_gpus'],'rccl',d['rccl_ranks'],'strong',d.get('strong'),'step_timing',d.get('step_timing'), d.get('gather'))
except Exception as e: print('g8 parse failed',e)
PY
tail -3 $O/bench_g8.err
