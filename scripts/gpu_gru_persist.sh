#!/bin/bash
# GPU box: persistent GRU tests, stamps, then C4 / C5 bench with MG_TUNE variants in the same call.
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "gru_persistent" > gpurun_out/gp_tests.log 2>&1
rc=$?; tail -n 25 gpurun_out/gp_tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit $rc; fi
if [ -f morgana_amd/libmorgana_hip_diag.so ]; then
  MG_TUNE=2:0 timeout -k 10 200 python scripts/stamps_gru.py 2>&1 | grep -v amdgpu.ids
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
fi
for cfg in ${CFGS:-c4 c5}; do
  for p in ${VARIANTS:-2:0 2:1 2:0 2:1}; do
    MG_TUNE=$p timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/gp_${cfg}_${p}.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out"; exit $rc; fi
    echo "$cfg MG_TUNE=$p: $(tail -n 1 gpurun_out/gp_${cfg}_${p}.log | grep -o '"ms_per_step": [0-9.]*')"
  done
done
exit 0
