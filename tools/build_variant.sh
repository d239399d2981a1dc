#!/bin/bash
# A/B builds of the HIP library: tools/build_variant.sh <name> <extra hipcc flags...>  ->  embree_amd/lib/variant_<name>.so  (use with MI355_LIB=...)
set -e
N=$1; shift
D=embree_amd/lib; C=embree_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w"
for s in build.hip trace.hip trace_fptr.hip shard.hip rtcore_api.cpp; do /opt/rocm/bin/hipcc -x hip $F "$@" -c $C/$s -o $D/v_${N}_$s.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,now -o $D/variant_$N.so $D/v_${N}_build.hip.o $D/v_${N}_trace.hip.o $D/v_${N}_trace_fptr.hip.o $D/v_${N}_shard.hip.o $D/v_${N}_rtcore_api.cpp.o -ldl
rm -f $D/v_${N}_*.o; ls -la $D/variant_$N.so
