#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace of the timed steps only (no roofline / CPU baseline legs), so that the
# per-step timeline (kernel order, durations, gaps) can be read off the trace CSV.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-trace}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/$TAG" -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS} > "$ROOT/gpurun_out/$TAG.log" 2>&1
echo "exit $?"
cd "$ROOT"
find gpurun_out/$TAG -name "*kernel_trace.csv" | head
