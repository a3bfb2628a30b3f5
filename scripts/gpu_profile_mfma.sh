#!/bin/bash
# On the GPU box (through gpurun): the matrix pipe's busy cycles per kernel of the default bench -
# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, the program directly behind `--`, counters in a run of their own (no trace
# domains).  usage: scripts/gpu_profile_mfma.sh <tag>   [BENCH_ARGS / MORGANA_* in the environment select the leg, as gpu_profile.sh]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_mfma" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-compare ${BENCH_ARGS} > "$ROOT/gpurun_out/prof_${TAG}_mfma.log" 2>&1
rc=$?
echo "[prof_${TAG}_mfma] exit $rc"
cd "$ROOT"
python3 scripts/summarise_mfma.py "$TAG" | head -30
exit $rc
