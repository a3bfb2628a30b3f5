#!/bin/bash
# Kernel trace of the default bench under an environment switch: usage gpu_trace_env.sh <tag> [VAR=VALUE ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${TAG}" -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-compare > "$ROOT/gpurun_out/${TAG}.log" 2>&1
echo "exit $?"
cd "$ROOT"
python3 scripts/prof_summary.py gpurun_out/${TAG} 10 > gpurun_out/${TAG}_summary.txt 2>&1
head -12 gpurun_out/${TAG}_summary.txt | cut -c1-150
