#!/bin/bash
# Same-box A/B of an environment switch on other bench configs: usage gpu_env_ab.sh "<VAR=VALUE ...> of the B leg" <config> [<config> ...]
# (three interleaved rounds per config; ms_per_step of both legs)
mkdir -p gpurun_out
B_ENV="$1"; shift
for cfg in "$@"; do
  for i in 1 2 3; do
    timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/env_a_${cfg}_$i.log 2>&1 || exit 1
    env $B_ENV timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/env_b_${cfg}_$i.log 2>&1 || exit 1
  done
  echo "$cfg: $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/env_a_${cfg}_*.log | grep -o '[0-9.]*$' | tr '\n' ' ') <- default | $(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/env_b_${cfg}_*.log | grep -o '[0-9.]*$' | tr '\n' ' ') <- $B_ENV"
done
