#!/bin/bash
# Static instruction mix of one kernel:  tools/isa_count.sh build.hip small_build [extra hipcc flags]  ->  VALU / SALU / LDS / VMEM counts (+ canonicalising v_max x,x)
SRC=$1; PAT=$2; shift 2
S=$(mktemp /tmp/isa_XXXX.s)
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w --cuda-device-only "$@" -S embree_amd/csrc/$SRC -o $S || exit 1
awk -v pat="$PAT" '$0 ~ "^_Z[A-Za-z0-9_]*"pat"[A-Za-z0-9_]*:" {f=1} f {print} f && /s_endpgm/ {exit}' $S > $S.k
echo "valu $(grep -cE '^\s*v_' $S.k) salu $(grep -cE '^\s*s_' $S.k) lds $(grep -cE '^\s*ds_' $S.k) vmem $(grep -cE '^\s+(global|buffer|scratch|flat)_' $S.k) canon $(grep -cE 'v_max_f32(_e32|_e64)? (v[0-9]+), (v[0-9]+), \3$' $S.k) b64shift $(grep -cE 'v_(lshl|lshr|ashr)rev_[bi]64' $S.k) rcp $(grep -cE 'v_rcp|v_div_' $S.k)"
grep -E "\.vgpr_count|\.sgpr_count" $S | head -0
rm -f $S $S.k
