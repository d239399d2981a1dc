#!/bin/bash
O=gpurun_out/r03k; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q --tb=short -k "coherent" -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|^E " $O/pytest.log | head -40
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err; tail -3 $O/configs.err; cat $O/configs.md


