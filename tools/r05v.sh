#!/bin/bash
# round 5: the whole GPU suite on the current tree
O=gpurun_out/r05v; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 > $O/pytest.log; cat $O/pytest.log | cut -c1-200
