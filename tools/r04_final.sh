#!/bin/bash
# end-of-round measurements: the driver's bench command, the default one, kernel stats + PMC of the same kernel source, the all-configs table
O=gpurun_out/r04z; mkdir -p $O
# (PMC first: the bench line only takes counters collected for exactly this kernel source)
bash tools/profile_round.sh r04 2>&1 | tail -3
cp gpurun_out/r04_pmc_trace.json profiles/pmc_bench_latest.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver cmd) rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err
# counters of the largest kernel of a commit (three commits per pass; the first one is dropped by the summary)
tools/pmc_run.sh gpurun_out/r04_pmc_sb python $PWD/tests/gpu_build_only.py "" 3 > gpurun_out/r04_pmc_sb.log 2>&1
python tools/pmc_summary.py gpurun_out/r04_pmc_sb "small_build" gpurun_out/r04_pmc_small_build > gpurun_out/r04_pmc_sb_summary.log 2>&1
# launch-by-launch timeline of a default and of a HIGH commit
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/r04_tl_raw -- python $OLDPWD/tests/gpu_build_only.py "" 4 > /dev/null 2>&1 )
python tools/ktimeline.py gpurun_out/r04_tl_raw v > $O/timeline_medium.txt 2>&1; rm -rf gpurun_out/r04_tl_raw
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/r04_tl_raw -- python $OLDPWD/tests/gpu_build_only.py "" 3 2 > /dev/null 2>&1 )
python tools/ktimeline.py gpurun_out/r04_tl_raw > $O/timeline_high.txt 2>&1; rm -rf gpurun_out/r04_tl_raw
tail -32 $O/timeline_medium.txt
python - <<'PY'
import json
for f in ('bench_driver','bench_default'):
    d=json.load(open('gpurun_out/r04z/%s.json'%f))
    print(f,'value',d['value'],'serial',d.get('serial',{}).get('value'),'build',d['build']['gpu_build_ms'],d['build']['mprims_per_s_gpu'],'roof',d['roofline']['frac'],d['roofline'].get('hbm_counter_from_profile',{}).get('frac'), d['roofline'].get('valu_from_profile',{}).get('frac'), d['roofline']['address_rate']['frac'], 'e2e', d.get('end_to_end',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('parity_vs_reference'))
PY
