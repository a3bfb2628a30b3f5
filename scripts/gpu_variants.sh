#!/bin/bash
mkdir -p gpurun_out
for v in 0 1 2 3 4 5; do
  echo "=== variant $v"
  MG_VARIANT=$v timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "linear_kernels_bf16 or full_size" -p no:cacheprovider 2>&1 | tail -1
  MG_VARIANT=$v timeout -k 10 120 python scripts/kbench.py 10 fwd1,fwd2,dgrad2 2>&1 | grep -E "fwd|dgrad"
done
