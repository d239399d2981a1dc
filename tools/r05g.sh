#!/bin/bash
# round 5, GPU call 7: reference programs with the scheduler shim, device filter functions
O=gpurun_out/r05g; mkdir -p $O
(cd $O && LD_PRELOAD=$PWD/../../tools/segv_trace.so timeout 300 ../../tests/golden/_bin/ref_triangle_geometry --compare ../../tests/golden/models/triangle_geometry.exr -o tg.ppm > tg.log 2>&1; echo "triangle_geometry rc=$?"; tail -12 tg.log)
LD_PRELOAD=$PWD/tools/segv_trace.so timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --durations=6 2>&1 | tail -40 > $O/pytest5.log; cat $O/pytest5.log
timeout 90 python tests/gpu_knobs.py cur 2>&1 | grep -E "KNOBS|rror"
