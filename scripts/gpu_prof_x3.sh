#!/bin/bash
# On the GPU box (through gpurun): kernel trace of scripts/x3_smoke.py (the fused 'bf16x3' step beside the bf16 one), per-kernel summary.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x3}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_$TAG" -o "$TAG" -- python3 "$ROOT/scripts/x3_smoke.py" ${2:-256} ${3:-1000} > "$ROOT/gpurun_out/prof_$TAG.log" 2>&1
rc=$?
cd "$ROOT"
grep -E "graph step|loss rel|grad|calls" "gpurun_out/prof_$TAG.log" | cut -c1-200
python3 scripts/prof_summary.py "gpurun_out/prof_$TAG" 100 2>&1 | cut -c1-160 | head -16
exit $rc
