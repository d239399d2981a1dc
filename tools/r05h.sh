#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
(cd $O && LD_PRELOAD=$PWD/../../tools/segv_trace.so timeout 300 ../../tests/golden/_bin/ref_triangle_geometry --compare ../../tests/golden/models/triangle_geometry.exr -o tg.ppm > tg.log 2>&1; echo "triangle_geometry rc=$?"; tail -12 tg.log)
LD_PRELOAD=$PWD/tools/segv_trace.so timeout 120 python tests/gpu_devfilter.py > $O/devfilter.log 2>&1; echo "devfilter rc=$?"; tail -30 $O/devfilter.log
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "verify or triangle_geometry" 2>&1 | tail -30 > $O/pytest5.log; cat $O/pytest5.log
