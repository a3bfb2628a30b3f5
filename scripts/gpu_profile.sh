#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace/stats of the default bench, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC runs carry no trace domains).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r1}
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
run() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 "$secs" "$@" > "$ROOT/gpurun_out/$log" 2>&1
    local rc=$?
    echo "[$log] exit $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out - stopping"; exit $rc; fi
    return 0
}
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compare ${BENCH_ARGS}"
run 600 prof_${TAG}_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_stats" -- $BENCH
run 600 prof_${TAG}_fetch.log rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_fetch" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-compare ${BENCH_ARGS}
run 600 prof_${TAG}_write.log rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_write" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-compare ${BENCH_ARGS}
cd "$ROOT"
find gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write -name "*.csv" | head -20
du -sh gpurun_out
exit 0
