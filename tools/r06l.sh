#!/bin/bash
# round 6: the profiles -- kernel stats of the default bench command, PMC passes of the closest-hit kernel (-> profiles/pmc_bench_latest.json), PMC passes of small_build, commit timelines
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06l; mkdir -p $O
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
tools/pmc_run.sh gpurun_out/r06_pmc_build python $R/tests/gpu_build_only.py "" 6 > $O/pmc_build.log 2>&1
rm -f gpurun_out/r06_pmc_small_build.md; python tools/pmc_summary.py gpurun_out/r06_pmc_build "small_build" gpurun_out/r06_pmc_small_build > $O/pmc_small_summary.log 2>&1; tail -12 $O/pmc_small_summary.log
( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_m $R/$O/prof_h && rocprofv3 --kernel-trace --stats -d $R/$O/prof_m -o commit -- python $R/tests/gpu_build_only.py "" 6 > $R/$O/prof_m.log 2>&1; rocprofv3 --kernel-trace --stats -d $R/$O/prof_h -o commit -- python $R/tests/gpu_build_only.py "" 5 2 > $R/$O/prof_h.log 2>&1 )
python tools/ktimeline.py $O/prof_m v > gpurun_out/r06_commit_timeline_medium.txt 2>&1; tail -32 gpurun_out/r06_commit_timeline_medium.txt
python tools/ktimeline.py $O/prof_h v > gpurun_out/r06_commit_timeline_high.txt 2>&1; tail -12 gpurun_out/r06_commit_timeline_high.txt | head -8
ls gpurun_out | head -30
