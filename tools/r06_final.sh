#!/bin/bash
# round 6, last session: the final tree -- the GPU suite three times over (does anything flake?), the driver's bench command, the default bench under rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06_final; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=12 > $O/suite_1.log 2>&1; echo "suite 1 rc=$?" | tee -a $O/suite_rc.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/kprof -o bench -- python $R/bench.py --no-cpu --sustain 0 > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err )
python tools/kstats.py $O/kprof > $O/bench_kernel_stats.md 2>&1
rm -rf $O/kprof
for i in 2 3; do timeout 1500 python -m pytest tests -m gpu -q -x > $O/suite_$i.log 2>&1; echo "suite $i rc=$?" | tee -a $O/suite_rc.log; tail -1 $O/suite_$i.log; done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
grep -a "passed\|failed" $O/suite_*.log | tail -5
tail -c 400 $O/bench_driver_cmd.json
