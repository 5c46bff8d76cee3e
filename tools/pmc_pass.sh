#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass (its own run, no trace options) of a short
# bench.py invocation, summarised per kernel.
#   gpurun -- 'bash tools/pmc_pass.sh "SQ_WAVES SQ_WAVE_CYCLES ..." [bench.py arguments]'
set -u
CTRS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_pass
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d "$O" -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$O/err.txt" || { tail -5 "$O/err.txt"; exit 1; }
python3 "$R/tools/pmc_summary.py" "$O"
