#!/bin/bash
# Kernel trace of bench.py for another workload: usage gpu_trace_cfg.sh <tag> <bench args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/$TAG" -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline "$@" > "$ROOT/gpurun_out/$TAG.log" 2>&1
echo "exit $?"
tail -1 "$ROOT/gpurun_out/$TAG.log" | cut -c1-200
