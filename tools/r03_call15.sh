#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
for i in 1 2 3; do python tests/gpu_debug2.py "" 2>&1 | grep -E "differing|info"; done
timeout 1200 python -m pytest tests -m gpu -q --tb=short --maxfail=15 --deselect tests/test_gpu_round3.py::test_shadow16m_whole_job_vs_reference_prefix -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/pytest.log | head -40
timeout 300 python tests/gpu_perf.py --tag tfar0 --reps 8 2>&1 | grep -E "PERF|rror" | tee -a $O/sweep.log
