#!/bin/bash
O=gpurun_out/r05x; mkdir -p $O
MI355_LIB=$PWD/embree_amd/lib/variant_prof.so timeout 120 python tests/gpu_segclk.py 2>&1 | grep SEGCLK > $O/segclk.log; cat $O/segclk.log
