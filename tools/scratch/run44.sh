timeout 900 python bench.py > gpurun_out/bench5.json 2> gpurun_out/bench5.err; tail -2 gpurun_out/bench5.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof44 -o r44 -- python /root/repo/bench.py --no-cpu > /root/repo/gpurun_out/bench5_prof.json 2> /root/repo/gpurun_out/bench5_prof.err
cd /root/repo
python tools/kstats.py gpurun_out/prof44 > gpurun_out/prof44_kstats.md
cat gpurun_out/bench5.json; head -30 gpurun_out/prof44_kstats.md
