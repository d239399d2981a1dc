#!/bin/bash
# round 5, GPU call 6: the reference's programs again (stride fix, crash trace), build flags A/B, new tests
O=gpurun_out/r05f; mkdir -p $O
(cd $O && LD_PRELOAD=$PWD/../../tools/segv_trace.so timeout 120 ../../tests/golden/_bin/ref_triangle_geometry --compare ../../tests/golden/models/triangle_geometry.exr -o tg.ppm > tg.log 2>&1; echo "triangle_geometry rc=$?"; tail -25 tg.log)
(cd $O && LD_PRELOAD=$PWD/../../tools/segv_trace.so timeout 120 ../../tests/golden/_bin/ref_triangle_geometry --threads 1 -o tg1.ppm > tg1.log 2>&1; echo "triangle_geometry --threads 1 rc=$?"; tail -6 tg1.log)
for v in "" bnoslp blds bboth; do
  echo "== build variant '$v'" >> $O/build_ab.log
  MI355_LIB=${v:+embree_amd/lib/variant_$v.so} TREEHASH=1 timeout 120 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/build_ab.log
  MI355_LIB=${v:+embree_amd/lib/variant_$v.so} timeout 120 python tests/gpu_build_only.py "" 6 2 2>&1 | grep -E "BUILD|rror|fault" >> $O/build_ab.log
done
cat $O/build_ab.log | cut -c1-230
V=tests/golden/_bin/ref_verify
for pat in ".*triangle_split_epsilon.*" ".*interpolate.*triangle.*" ".*regression_static" ".*sphere_filter_multi_hit_tests.*" ".*buffer_stride.quads"; do
  timeout 150 $V --no-colors --sequential --intensity 0.2 --run "$pat" > $O/v.tmp 2>&1; echo "== $pat rc=$?" >> $O/verify_more.txt
  grep -E "PASSED|FAILED|rror|Tests" $O/v.tmp | tail -6 >> $O/verify_more.txt
done
cat $O/verify_more.txt | cut -c1-170
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -x -q --durations=5 2>&1 | tail -14 > $O/pytest5.log; cat $O/pytest5.log
