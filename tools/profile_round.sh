#!/bin/bash
# Collects what bench.py's roofline section and profiles/ need, on the GPU box (run from the repo root):
#   tools/profile_round.sh <tag>      ->  gpurun_out/<tag>_kstats.md, gpurun_out/<tag>_pmc (+ profiles/pmc_bench_latest.json, profiles/<tag>_pmc_trace.md)
# 1. rocprofv3 --kernel-trace --stats of the default bench command (kernel durations: must agree with roofline.kernel_ms_avg)
# 2. the PMC passes of tools/pmc_run.sh (counters in their own runs, never mixed with tracing), summarised for the closest-hit kernel together with
#    the hash of the kernel source they were collected for
T=${1:-r03}
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o ${T} -- python $R/bench.py --no-cpu --sustain 0 > $R/gpurun_out/${T}_prof_bench.json 2> $R/gpurun_out/${T}_prof_bench.err )
python tools/kstats.py gpurun_out/${T}_prof > gpurun_out/${T}_kstats.md 2>&1
tools/pmc_run.sh gpurun_out/${T}_pmc python $R/bench.py --steps 8 --warmup 2 --spin-up 0.05 --no-cpu --sustain 0 --inprocess-gpus 1 --streams 1 --pipeline-streams 0 > gpurun_out/${T}_pmc.log 2>&1
rm -f gpurun_out/${T}_pmc_trace.md
python tools/pmc_summary.py gpurun_out/${T}_pmc "trace_kernel_q<false, false, false, false, 0, false>" gpurun_out/${T}_pmc_trace > gpurun_out/${T}_pmc_summary.log 2>&1
tail -5 gpurun_out/${T}_pmc_summary.log
