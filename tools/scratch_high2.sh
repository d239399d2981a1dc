#!/bin/bash
# per-level times of the spatial kernels for library variants: tools/scratch_high2.sh v1 v2 ...
R=$PWD
for v in "$@"; do
  ( cd /tmp && export TMPDIR=/tmp && MI355_LIB=$R/embree_amd/lib/variant_$v.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/high_$v -o high -- python $R/tests/gpu_build_only.py "" 2 2 > $R/gpurun_out/high_$v.log 2>&1 )
  grep BUILD gpurun_out/high_$v.log
  python - <<EOF
import sqlite3,glob
f=glob.glob('gpurun_out/high_$v/*.db')[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(db.execute(f"select s.kernel_name,(d.end-d.start)/1e3,d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
seq=[(n,us) for n,us,st in rows]
start=[i for i,(n,_) in enumerate(seq) if 'build_begin' in n][-1]
out=[]
for n,us in seq[start:]:
    if 'spatial_partition' in n: out.append("%.0f"%us)
print("$v spatial_partition per level:", ' '.join(out[:8]))
out=[]
for n,us in seq[start:]:
    if 'spatial_bin' in n: out.append("%.0f"%us)
print("$v spatial_bin per level:", ' '.join(out[:16]))
EOF
done
