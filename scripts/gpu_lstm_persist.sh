#!/bin/bash
# GPU box: LSTM tests, then the LSTM acoustic model bench with the persistent recurrence on / off in the same call.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "lstm" > gpurun_out/lp_tests.log 2>&1
rc=$?; tail -n 25 gpurun_out/lp_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
for p in 1 0 1; do
  MORGANA_PERSISTENT=$p timeout -k 10 400 python bench.py --config lstm --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/lp_lstm_${p}.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out"; exit $rc; fi
  echo "lstm persistent=$p: $(tail -n 1 gpurun_out/lp_lstm_${p}.log | grep -o '"ms_per_step": [0-9.]*')"
done
exit 0
