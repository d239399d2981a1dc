#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O
for args in "--tag outlier16" "--config top_split_cell=0.03125 --tag outlier32" "--config top_split_cell=0.125 --tag outlier8" "--config top_splits=0 --tag nosplit" "--powerplant --tag pp"; do
  timeout 300 python tests/gpu_perf.py $args --reps 6 2>&1 | grep -E "PERF|rror" | tee -a $O/sweep.log
done
python tests/gpu_build_only.py "" 6 2>&1 | tee $O/build_default.log
python tests/gpu_build_only.py "top_splits=0" 6 2>&1 | tee $O/build_nosplit.log
PP=1 python tests/gpu_build_only.py "" 4 2>&1 | tee $O/build_pp.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short --maxfail=15 --deselect tests/test_gpu_round3.py::test_shadow16m_whole_job_vs_reference_prefix -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/pytest.log | head -40
