#!/bin/bash
# Same-box A/B of a tuning switch on other bench configs: usage gpu_tune_ab_cfg.sh <MG_TUNE of the B leg> <config> [<config> ...]
mkdir -p gpurun_out
B_TUNE="$1"; shift
for cfg in "$@"; do
  for i in 1 2 3; do
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/tune_a_${cfg}_$i.log 2>&1 || exit 1
    MG_TUNE=$B_TUNE timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/tune_b_${cfg}_$i.log 2>&1 || exit 1
  done
  echo "$cfg: $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/tune_a_${cfg}_*.log | grep -o '[0-9.]*$' | tr '\n' ' ') <- default | $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/tune_b_${cfg}_*.log | grep -o '[0-9.]*$' | tr '\n' ' ') <- MG_TUNE=$B_TUNE"
done
