#!/bin/bash
# round 6, session 5, final tree (+ small_build largest first): the driver's bench command, commit timelines, counter traffic, PMC of small_build, kernel stats
# three qualities, counter traffic of a commit, PMC of small_build, kernel stats of the bench command
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06zr; mkdir -p $O; rm -rf $O/*
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
for Q in "m 6" "l 6 0" "h 5 2"; do set -- $Q; T=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_$T && rocprofv3 --kernel-trace --stats -d $R/$O/prof_$T -o commit -- python $R/tests/gpu_build_only.py "" "$@" > $R/$O/prof_$T.log 2>&1 )
  python tools/ktimeline.py $O/prof_$T v > $O/commit_timeline_$T.txt 2>&1
  rm -rf $O/prof_$T
done
tools/pmc_run.sh $O/pmc_build python $R/tests/gpu_build_only.py "" 6 > $O/pmc_build.log 2>&1
python tools/pmc_summary.py $O/pmc_build "small_build" $O/pmc_small_build > $O/pmc_small_summary.log 2>&1
python tools/commit_traffic.py $O/pmc_build $O/commit_traffic | tail -3
rm -rf $O/pmc_build/p*/*.db
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/kprof -o bench -- python $R/bench.py --no-cpu --sustain 0 > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err )
python tools/kstats.py $O/kprof > $O/bench_kernel_stats.md 2>&1
rm -rf $O/kprof
tail -3 $O/bench_driver_cmd.json | cut -c1-600
