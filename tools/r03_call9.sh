#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
for ml in 32 40 48 56; do
  echo "== MI355_PACKET_MIN_LANES=$ml" | tee -a $O/configs.md
  MI355_PACKET_MIN_LANES=$ml timeout 600 python tests/gpu_configs.py 2> $O/configs.err | grep -E "COHERENT|coherent primary|shard" | tee -a $O/configs.md
done
