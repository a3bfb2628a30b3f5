#!/bin/bash
# GPU box: A/B of two builds of the library on bench configs within one call.  usage: gpu_ab_lib.sh <libA> <libB> <configs...>
mkdir -p gpurun_out
a=$1; b=$2; shift 2
for cfg in "$@"; do
  for lib in $a $b $a $b; do
    MORGANA_HIP_LIB=$PWD/$lib timeout -k 10 400 python bench.py --config $cfg --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ablib.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out"; exit $rc; fi
    echo "$cfg $lib: $(tail -n 1 gpurun_out/ablib.log | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' ')"
  done
done
exit 0
