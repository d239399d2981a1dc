#!/bin/bash
# the whole GPU suite (the heavy configs[3] whole-job test last) -- what the driver runs at round end
O=gpurun_out/r03suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=10 -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/pytest.log | head -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
