#!/bin/bash
# Same-box sweep of MG_TUNE settings on the C2 bench: usage gpu_sweep.sh <setting> [<setting> ...]   ("-" = the default)
mkdir -p gpurun_out
for rep in 1 2; do
  for t in "$@"; do
    if [ "$t" = "-" ]; then tune=""; else tune="$t"; fi
    MG_TUNE=$tune timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline --no-compare > gpurun_out/sweep.log 2>&1 || exit 1
    echo "$t $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/sweep.log)"
  done
done
