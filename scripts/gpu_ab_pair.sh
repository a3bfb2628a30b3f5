#!/bin/bash
# Same-box A/B of the paired weight-gradient + dgrad grid (MG_TUNE=7:65 = the two launches): test, then the C2 bench three times each way.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "pair or l2tail or stack" > gpurun_out/ab_tests.log 2>&1
echo "tests exit $?"; tail -n 5 gpurun_out/ab_tests.log
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline --no-compare > gpurun_out/ab_pair_$i.log 2>&1 || exit 1
  MG_TUNE=7:65 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline --no-compare > gpurun_out/ab_two_$i.log 2>&1 || exit 1
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab_pair_*.log | tr '\n' ' '; echo " <- pair"
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab_two_*.log | tr '\n' ' '; echo " <- two launches"
