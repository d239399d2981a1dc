cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > /root/repo/gpurun_out/counters_list.txt 2>&1
cd /root/repo
export MI355_NUM_CURSORS=1 MI355_REFILL_MIN=32
tools/pmc_run.sh gpurun_out/pmc3 python /root/repo/tests/gpu_perf.py --reps 3 --tag pmc > gpurun_out/pmc3.log 2>&1
