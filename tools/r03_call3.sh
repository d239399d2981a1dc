#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O
for args in "--config top_splits=0 --tag nosplit" "--tag topsplit64k" "--config top_split_min=16384 --tag topsplit16k" "--config top_split_min=262144 --tag topsplit256k" "--config top_split_min=4096 --tag topsplit4k"; do
  timeout 300 python tests/gpu_perf.py $args --reps 6 2>&1 | grep -E "PERF|Error|error|rror" | tee -a $O/sweep.log
done
timeout 1200 python -m pytest tests -m gpu -q --tb=short --maxfail=15 --deselect tests/test_gpu_round3.py::test_shadow16m_whole_job_vs_reference_prefix -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | head -40
