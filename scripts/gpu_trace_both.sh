#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace of the timed C2 steps in both orders of operations (phone rate, frame rate),
# summarised per kernel over the last dispatches (the timed replays): usage gpu_trace_both.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-tb}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
export MORGANA_PHONE_RATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${TAG}_fr" -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-compare ${BENCH_ARGS} > "$ROOT/gpurun_out/${TAG}_fr.log" 2>&1
echo "frame rate: exit $?"
export MORGANA_PHONE_RATE=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${TAG}_pr" -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-compare ${BENCH_ARGS} > "$ROOT/gpurun_out/${TAG}_pr.log" 2>&1
echo "phone rate: exit $?"
cd "$ROOT"
python3 scripts/prof_summary.py gpurun_out/${TAG}_fr 10 > gpurun_out/${TAG}_fr_summary.txt 2>&1
python3 scripts/prof_summary.py gpurun_out/${TAG}_pr 10 > gpurun_out/${TAG}_pr_summary.txt 2>&1
head -16 gpurun_out/${TAG}_fr_summary.txt | cut -c1-130
head -16 gpurun_out/${TAG}_pr_summary.txt | cut -c1-130
