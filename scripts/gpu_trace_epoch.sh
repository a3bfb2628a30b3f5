#!/bin/bash
# On the GPU box (through gpurun): kernel trace of scripts/train_epoch_trace.py and the per-step timeline of its last epoch.
# usage: bash scripts/gpu_trace_epoch.sh <tag> [precision] [n_batches] [steps per graph]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-te}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_$TAG" -o "$TAG" -- python3 "$ROOT/scripts/train_epoch_trace.py" ${2:-bf16} ${3:-52} ${4:-10} > "$ROOT/gpurun_out/prof_$TAG.log" 2>&1
rc=$?
cd "$ROOT"
grep "ms per step" "gpurun_out/prof_$TAG.log"
python3 scripts/train_epoch_gaps.py "gpurun_out/prof_$TAG/${TAG}_results.db" ${3:-52} 2>&1 | cut -c1-400
exit $rc
