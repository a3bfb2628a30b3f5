#!/bin/bash
# Run on the GPU box (through gpurun): GPU parity tests, smoke, a short bench.  Stops after a timed-out step.
mkdir -p gpurun_out
run() {  # run <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
    local rc=$?
    echo "[$log] exit $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out - stopping"; exit $rc; fi
    return 0
}
run 900 tests.log python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS}
tail -n 40 gpurun_out/tests.log
run 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
tail -n 5 gpurun_out/smoke.log
run 600 bench.log python bench.py ${BENCH_ARGS}
tail -n 5 gpurun_out/bench.log
exit 0
