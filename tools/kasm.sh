#!/bin/bash
# tools/kasm.sh <out.s> [extra hipcc flags]: assembly of the default closest-hit kernel trace_kernel_q<false,false,false,false,false> + its static instruction mix
O=$1; shift
S=$(mktemp /tmp/kasm_XXXX.s)
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize -w --cuda-device-only "$@" -S embree_amd/csrc/trace.hip -o $S || exit 1
awk '/^_ZN12_GLOBAL__N_114trace_kernel_qILb0ELb0ELb0ELb0ELi0EEEvNS_9TraceArgsE:/ {f=1} f {print} f && /^; Occupancy/ {exit}' $S > $O
echo "valu $(grep -cE '^\s*v_' $O) salu $(grep -cE '^\s*s_' $O) lds $(grep -cE '^\s*ds_' $O) vmem $(grep -cE '^\s+(global|buffer|scratch|flat)_' $O) cndmask_vcc $(grep -c 'v_cndmask_b32_e32' $O) | $(grep -E 'NumVgprs|ScratchSize|Occupancy' $O | tr '\n' ' ')"
rm -f $S
