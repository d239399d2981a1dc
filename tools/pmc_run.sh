#!/bin/bash
# PMC passes for the trace kernel (run on the GPU box). Counters are collected in their own runs
# (no --kernel-trace/--stats mixed with --pmc), one pass per counter group.  Usage: tools/pmc_run.sh <outdir> <cmd...>
# (the command runs with cwd=/tmp: give script paths as /root/repo/...)
OUT=$1; shift
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/p$i -o pmc -- "$@" > $R/$OUT/p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
cd $R
find $OUT -name "*counter_collection.csv" | head
