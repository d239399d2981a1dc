#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short --maxfail=15 --deselect tests/test_gpu_round3.py::test_shadow16m_whole_job_vs_reference_prefix -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/pytest.log | head -40
python tests/gpu_build_only.py "" 6 2>&1 | tee $O/build_default.log
timeout 300 python tests/gpu_perf.py --tag final --reps 6 2>&1 | grep -E "PERF|rror" | tee -a $O/sweep.log
