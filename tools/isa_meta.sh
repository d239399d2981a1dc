#!/bin/bash
# Register / LDS / scratch budget of the kernels of one source file, from the compiler's own code-object metadata:
#   tools/isa_meta.sh build.hip|trace.hip [pattern]   ->  name  vgpr  agpr  sgpr  lds_bytes  scratch_bytes
SRC=${1:-trace.hip}; PAT=$2
S=$(mktemp /tmp/isa_XXXX.s)
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w --cuda-device-only -S embree_amd/csrc/$SRC -o $S || exit 1
python3 - "$S" "$PAT" <<'PY'
import sys, re, subprocess
txt = open(sys.argv[1]).read(); pat = sys.argv[2] if len(sys.argv) > 2 else ''
meta = txt[txt.index('amdhsa.kernels:'):] if 'amdhsa.kernels:' in txt else ''
for blk in re.split(r'\n  - ', meta)[1:]:
    f = dict(re.findall(r'\.(\w+):\s+(\S+)', blk))
    n = f.get('name', '?')
    n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
    if pat and pat not in n: continue
    print('%-100s vgpr %4s agpr %3s sgpr %4s lds %6s scratch %5s' % (n[:100], f.get('vgpr_count'), f.get('agpr_count'), f.get('sgpr_count'), f.get('group_segment_fixed_size'), f.get('private_segment_fixed_size')))
PY
rm -f $S
