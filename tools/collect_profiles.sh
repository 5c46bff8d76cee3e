#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of bench.py plus the HBM-traffic
# PMC passes (each in a run of its own), written under gpurun_out/<tag>/.
#   gpurun -- 'bash tools/collect_profiles.sh r03 [bench.py arguments, e.g. --workload sampling4d-L]'
set -u
TAG=${1:-r03}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- \
    python3 "$R/bench.py" --no-cpu-baseline "$@" > "$O/bench_under_rocprof.json" 2> "$O/stats.err" || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum"; do
    name=$(echo "$pass" | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$O/pmc_$name" -- \
        python3 "$R/bench.py" --steps 5 --warmup 2 --settle-ms 20 --no-cpu-baseline "$@" > /dev/null 2> "$O/pmc_$name.err" || exit 1
done
timeout -k 10 60 "$R/tools/calib_fetch" > "$O/calib.txt" 2>&1 || true
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- "$R/tools/calib_fetch" > /dev/null 2>&1 || true
# keep what the summaries are made of (kernel stats, mean counter values): the raw files are large
find "$O" -name "*kernel_trace.csv" -delete -o -name "*domain_stats.csv" -delete
python3 "$R/tools/pmc_reduce.py" "$O"
echo "profiles collected under $O"
