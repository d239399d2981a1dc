R=$PWD
mkdir -p gpurun_out/pmc24
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc24/p$i -o pmc -- python /root/repo/tests/gpu_perf.py --reps 1 > $R/gpurun_out/pmc24/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc24 2>&1 | tail -40
