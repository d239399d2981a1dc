#!/bin/bash
# A/B builds that differ in build.hip only: tools/build_variant.sh <name> <extra hipcc flags...>  ->  embree_amd/lib/variant_<name>.so (the other objects are the product's own; MI355_LIB=... loads it)
set -e
N=$1; shift
D=embree_amd/lib; C=embree_amd/csrc
python -c "from embree_amd import build; build.build()" > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w"   # (the flags of embree_amd/build.py for build.hip)
/opt/rocm/bin/hipcc -x hip $F "$@" -c $C/build.hip -o $D/v_${N}_build.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,now -o $D/variant_$N.so $D/v_${N}_build.hip.o $D/trace.hip.o $D/trace_fptr.hip.o $D/shard.hip.o $D/rtcore_api.cpp.o -ldl
rm -f $D/v_${N}_*.o; ls -la $D/variant_$N.so
