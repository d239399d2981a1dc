R=$PWD
mkdir -p gpurun_out/pmc40
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc40/p$i -o pmc -- python /root/repo/tests/gpu_perf.py --reps 1 > $R/gpurun_out/pmc40/p$i.log 2>&1
  echo "pass $i rc=$?"
done
