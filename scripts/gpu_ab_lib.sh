#!/bin/bash
# Same-box A/B of two builds of the library on the C2 bench: usage gpu_ab_lib.sh <path of the B library> [pytest -k expression]
# (MORGANA_HIP_LIB selects the library the package loads; build the B leg from an older source with the Makefile's object list)
mkdir -p gpurun_out
B_LIB=${1:?path of the B library}
if [ -n "$2" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$2" > gpurun_out/ab_tests.log 2>&1
  rc=$?; echo "tests exit $rc"; tail -n 15 gpurun_out/ab_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_a_$i.log 2>&1 || exit 1
  MORGANA_HIP_LIB=$B_LIB timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_b_$i.log 2>&1 || exit 1
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab_a_*.log | tr '\n' ' '; echo " <- this build (phone-rate step, frame-rate order)"
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/ab_b_*.log | tr '\n' ' '; echo " <- $B_LIB"
