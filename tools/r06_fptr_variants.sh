#!/bin/bash
# round 6, VERDICT r05 item 6: the calling kernels (FILT == 2, trace_fptr.hip) at -O3 with one suspect removed at a time.  Builds embree_amd/lib/variant_fp_<name>.so =
# the shipped objects + trace_fptr.hip recompiled with the given flags (run in the build container; the GPU side: tools/r06i.sh)
set -e
D=embree_amd/lib; C=embree_amd/csrc
F="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w -fno-slp-vectorize"
build() { N=$1; shift; /opt/rocm/bin/hipcc -x hip $F "$@" -c $C/trace_fptr.hip -o $D/v_$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,now -o $D/variant_fp_$N.so $D/build.hip.o $D/trace.hip.o $D/v_$N.o $D/shard.hip.o $D/rtcore_api.cpp.o -ldl; rm -f $D/v_$N.o; }
build o1 -O1
build o3 -O3
build o3_nosgpr -O3 -DMI355_SEL_SGPR=0
build o3_noscan -O3 -DMI355_PUSH_SCAN=0
build o3_noundef -O3 -DMI355_NO_UNDEF
build o3_nopre -O3 -DMI355_TRI_PREFETCH=0
build o3_all -O3 -DMI355_SEL_SGPR=0 -DMI355_PUSH_SCAN=0 -DMI355_NO_UNDEF -DMI355_TRI_PREFETCH=0
build o2_all -O2 -DMI355_SEL_SGPR=0 -DMI355_PUSH_SCAN=0 -DMI355_NO_UNDEF -DMI355_TRI_PREFETCH=0
ls -la $D/variant_fp_*.so
