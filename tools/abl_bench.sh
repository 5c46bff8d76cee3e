#!/bin/bash
# Runs ON THE GPU BOX: bench.py once per library build given on the command line (paths
# relative to the repo), printing step time and per-kernel medians.  WL / PTS choose workload / points.
for lib in "$@"; do
  OPUS_MSMV_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload ${WL:-L-fwdbwd} --points ${PTS:-uniform} --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline > gpurun_out/abl_tmp.json 2>/dev/null || { echo "FAILED $lib"; continue; }
  python - "$lib" <<'PY'
import json,sys
d=json.load(open("gpurun_out/abl_tmp.json"))
print(sys.argv[1].split("/")[-1].ljust(14), d["config"]["name"], d["config"]["points"], "step %.1f"%(d["ms_per_step"]*1e3), " ".join("%s=%.1f"%(n.replace("msmv_",""),v["p50_us"]) for n,v in d["kernels"].items()), flush=True)
PY
done
