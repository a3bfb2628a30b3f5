#!/bin/bash
# Same-box A/B/C... of tuning switches on the C2 bench: usage gpu_abn.sh "<pytest -k expression or empty>" <MG_TUNE leg> [<MG_TUNE leg> ...]
# (a leg is key:value[,key:value]; "-" is the default build state).  Three interleaved rounds, ms_per_step of every leg.
mkdir -p gpurun_out
K="$1"; shift
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > gpurun_out/ab_tests.log 2>&1
  rc=$?; echo "tests exit $rc"; tail -n 15 gpurun_out/ab_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
EXTRA=${MG_BENCH_EXTRA:-}
NOCMP=--no-compare; [ -n "$MG_BENCH_COMPARE" ] && NOCMP=   # MG_BENCH_COMPARE=1: time the frame-rate order as well
for i in 1 2 3; do
  for leg in "$@"; do
    tag=$(echo "$leg" | tr ':,' '__')
    if [ "$leg" = "-" ]; then
      timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline $NOCMP $EXTRA > gpurun_out/abn_${tag}_$i.log 2>&1 || exit 1
    else
      MG_TUNE=$leg timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-roofline --no-cpu-baseline $NOCMP $EXTRA > gpurun_out/abn_${tag}_$i.log 2>&1 || exit 1
    fi
  done
done
for leg in "$@"; do
  tag=$(echo "$leg" | tr ':,' '__')
  echo "$(grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/abn_${tag}_*.log | head -3 | tr '\n' ' ') | frame: $(grep -h -o '"frame_rate_order": {"ms_per_step": [0-9.]*' gpurun_out/abn_${tag}_*.log | grep -o '[0-9.]*$' | tr '\n' ' ')  <- $leg"
done
