#!/bin/bash
# GPU box: persistent GRU tests, then C4 bench with persistent on/off in the same call.
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "gru_persistent" > gpurun_out/gp_tests.log 2>&1
rc=$?; tail -n 25 gpurun_out/gp_tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit $rc; fi
for cfg in c4 c5; do
  for p in 2:0 2:1 2:0 2:1; do
    MG_TUNE=$p timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/gp_${cfg}_${p}.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out"; exit $rc; fi
    echo "$cfg MG_TUNE=$p (2:0 L2-local where possible, 2:1 always write-through): $(tail -n 1 gpurun_out/gp_${cfg}_${p}.log | grep -o '"ms_per_step": [0-9.]*')"
  done
done
exit 0
