#!/bin/bash
# per-level times of top_bin / top_partition of a MEDIUM commit for library variants (or "intree")
R=$PWD
for v in "$@"; do
  L=$R/embree_amd/lib/variant_$v.so; [ $v = intree ] && L=$R/embree_amd/lib/libembree4_mi355.so
  ( cd /tmp && export TMPDIR=/tmp && MI355_LIB=$L MI355_BUILD_STEPWISE=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lv_$v -o lv -- python $R/tests/gpu_build_only.py "" 2 > $R/gpurun_out/lv_$v.log 2>&1 )
  python - <<EOF
import sqlite3,glob
f=glob.glob('gpurun_out/lv_$v/*.db')[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(db.execute(f"select s.kernel_name,(d.end-d.start)/1e3,d.start from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
seq=[(n,us) for n,us,st in rows]
start=[i for i,(n,_) in enumerate(seq) if 'primref_gen' in n][-1]
for k in ('top_bin','top_partition','top_setup','top_split','top_emit','wide_plan','wide_emit','wide_scan'):
    print("$v", k, ' '.join("%.0f"%us for n,us in seq[start:] if k in n))
EOF
done
