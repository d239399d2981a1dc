#!/bin/bash
# round 6, final tree: kernel stats of the default bench command + PMC passes of the closest-hit kernel (-> profiles/pmc_bench_latest.json), the driver's bench command, commit timelines of the three qualities, PMC of small_build
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06v; mkdir -p $O; rm -rf $O/*
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -4 $O/profile_round.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
for Q in "m 6" "l 6 0" "h 5 2"; do set -- $Q; T=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_$T && rocprofv3 --kernel-trace --stats -d $R/$O/prof_$T -o commit -- python $R/tests/gpu_build_only.py "" "$@" > $R/$O/prof_$T.log 2>&1 )
  python tools/ktimeline.py $O/prof_$T v > $O/commit_timeline_$T.txt 2>&1
done
tail -30 $O/commit_timeline_m.txt | head -16
tools/pmc_run.sh $O/pmc_build python $R/tests/gpu_build_only.py "" 6 > $O/pmc_build.log 2>&1
python tools/pmc_summary.py $O/pmc_build "small_build" $O/pmc_small_build > $O/pmc_small_summary.log 2>&1; tail -3 $O/pmc_small_summary.log
