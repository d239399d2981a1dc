#!/bin/bash
O=gpurun_out/r03b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short --maxfail=15 --deselect tests/test_gpu_round3.py::test_shadow16m_whole_job_vs_reference_prefix -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | head -40
for args in "--tag medium" "--high --config presplits=1 --tag presplit20" "--high --config presplits=1,max_spatial_split_replications=1.02 --tag presplit2" "--high --config presplits=1,max_spatial_split_replications=1.005 --tag presplit05" "--high --tag high" ; do
  timeout 300 python tests/gpu_perf.py $args --reps 6 2>&1 | grep -E "PERF|Error|error" | tee -a $O/sweep.log
done
