#!/bin/bash
# GPU box: A/B of tuning legs on another workload of bench.py: usage gpu_cfg_ab.sh <config> <leg> [<leg> ...]  (leg: key:value or -)
mkdir -p gpurun_out
CFG=$1; shift
for i in 1 2; do
  for leg in "$@"; do
    tag=$(echo "$leg" | tr ':,' '__')
    if [ "$leg" = "-" ]; then
      timeout -k 10 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/cab_${CFG}_${tag}_$i.log 2>&1 || exit 1
    else
      MG_TUNE=$leg timeout -k 10 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/cab_${CFG}_${tag}_$i.log 2>&1 || exit 1
    fi
    echo "$CFG $leg: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/cab_${CFG}_${tag}_$i.log | head -1)"
  done
done
