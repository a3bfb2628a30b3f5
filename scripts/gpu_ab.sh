#!/bin/bash
# Same-box A/B of one tuning switch on the C2 bench: usage gpu_ab.sh <MG_TUNE value of the B leg> [pytest -k expression]
# (e.g. 7:65 = the weight gradient and the dgrad as two launches, 7:66 = the phone-rate front and the first GEMM as separate launches)
mkdir -p gpurun_out
B_TUNE=${1:?MG_TUNE of the B leg}
if [ -n "$2" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$2" > gpurun_out/ab_tests.log 2>&1
  rc=$?; echo "tests exit $rc"; tail -n 15 gpurun_out/ab_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline --no-compare > gpurun_out/ab_a_$i.log 2>&1 || exit 1
  MG_TUNE=$B_TUNE timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline --no-compare > gpurun_out/ab_b_$i.log 2>&1 || exit 1
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab_a_*.log | tr '\n' ' '; echo " <- default"
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab_b_*.log | tr '\n' ' '; echo " <- MG_TUNE=$B_TUNE"
