#!/bin/bash
# PMC pass over scripts/kbench.py: usage gpu_pmc.sh <tag> "<counters>" [kbench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; CNT=$2; shift 2
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --pmc $CNT --output-format csv -d "$ROOT/gpurun_out/pmc_${TAG}" -- python3 $ROOT/scripts/kbench.py "$@" > "$ROOT/gpurun_out/pmc_${TAG}.log" 2>&1
echo "exit $?"
tail -8 "$ROOT/gpurun_out/pmc_${TAG}.log"
