#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of one bench.py invocation, printed as a table.
#   gpurun -- 'bash tools/kstats.sh [bench.py arguments]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/kstats
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- \
    python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$O/bench.json" 2> "$O/err.txt" || { tail -5 "$O/err.txt"; exit 1; }
f=$(find "$O" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "msmv" in n:
        print("%-90s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
