#!/bin/bash
# GPU box: GRU tests, then C4 / C5 bench with the bf16 and the fp32 recurrence in the same call (devices differ between calls).
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "gru or rnn" > gpurun_out/gru_tests.log 2>&1
rc=$?; tail -n 30 gpurun_out/gru_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for cfg in c4 c5; do
  for rec in bf16 fp32 bf16 fp32; do
    MORGANA_RECURRENCE=$rec timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_${cfg}_${rec}.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out"; exit $rc; fi
    echo "$cfg $rec: $(tail -n 1 gpurun_out/ab_${cfg}_${rec}.log | cut -c1-200)"
  done
done
exit 0
