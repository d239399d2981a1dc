#!/bin/bash
# background load the round-end driver also applies: a rocm-smi sampler polling the GPU while the benchmark runs (usage: tools/smi_poll.sh <seconds>)
end=$((SECONDS + ${1:-30}))
while [ $SECONDS -lt $end ]; do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; sleep 0.2; done
