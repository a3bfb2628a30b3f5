#!/bin/bash
# rocprofv3 kernel statistics of the secondary workloads (C4 GRU, LSTM acoustic model) and the in-kernel stamp report.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
for cfg in ${CFGS:-c4 lstm f0gru}; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r1_$cfg" -- python3 $ROOT/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$ROOT/gpurun_out/prof_r1_$cfg.log" 2>&1
    rc=$?
    echo "[$cfg] exit $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit $rc; fi
done
cd "$ROOT"
timeout -k 10 300 python scripts/stamps.py > gpurun_out/stamps_r1.txt 2>&1
echo "[stamps] exit $?"
timeout -k 10 300 python scripts/stamps_gru.py > gpurun_out/stamps_gru_r1.txt 2>&1
echo "[stamps gru] exit $?"
tail -1 gpurun_out/prof_r1_c4.log 2>/dev/null | cut -c1-250
tail -1 gpurun_out/prof_r1_lstm.log 2>/dev/null | cut -c1-250
