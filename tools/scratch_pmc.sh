#!/bin/bash
# quick PMC A/B: SQ instruction counters for two refill thresholds
R=$PWD
cd /tmp && export TMPDIR=/tmp
for g in 32 16; do
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"; do
    n=$(echo $grp | cut -c1-12 | tr ' ' '_')
    MI355_REFILL_MIN=$g timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcq_g${g}_$n -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --pipeline-streams 0 > $R/gpurun_out/pmcq_g${g}_$n.log 2>&1
  done
done
cd $R
python - <<'P'
import csv, glob, collections
for g in (32, 16):
    agg = collections.defaultdict(list)
    for p in glob.glob("gpurun_out/pmcq_g%d_*/**/*counter_collection.csv" % g, recursive=True):
        for r in csv.DictReader(open(p)):
            if "trace_kernel_q<false, false, false, false>" in r["Kernel_Name"] and int(r["Grid_Size"]) == 327680:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(g, {k: "%.4g" % (sum(v[2:]) / max(1, len(v) - 2)) for k, v in sorted(agg.items())}, "launches", {k: len(v) for k, v in agg.items()}.get("SQ_INSTS_VALU"))
P
