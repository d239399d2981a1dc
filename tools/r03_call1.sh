#!/bin/bash
# GPU call 1 of round 3: the whole GPU suite (all failures, not just the first), then the collapse sweep on the full crown stand-in
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short --maxfail=15 --deselect tests/test_gpu_round3.py::test_shadow16m_whole_job_vs_reference_prefix > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -30 $O/pytest.log
for cfg in "collapse=greedy" "" "dp_tri_cost=0.3" "dp_tri_cost=0.7" "dp_tri_cost=1.0" "min_leaf=1" "min_leaf=1,dp_tri_cost=0.3" "dp_tri_cost=0.35,dp_node_cost=1" ; do
  timeout 300 python tests/gpu_perf.py --config "$cfg" --tag "sweep" --reps 6 2>&1 | grep -E "PERF|lane-iter|Error|error" | tee -a $O/sweep.log
done
