#!/bin/bash
# Runs ON THE GPU BOX: bench.py once per value of an environment switch of the library, printing step
# time and per-kernel medians.   VAR=OPUS_MSMV_DFEAT_H16 VALS="0 1" WL=decoder-L PTS=uniform bash tools/ab_env.sh
for v in ${VALS:-0 1}; do
  env ${VAR}=$v timeout -k 10 200 python bench.py --workload ${WL:-L-fwdbwd} --points ${PTS:-uniform} --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null || { echo "FAILED $v"; continue; }
  python - "$VAR=$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1].ljust(24), d["config"]["name"], d["config"]["points"], "step %.1f"%(d["ms_per_step"]*1e3), " ".join("%s=%.1f"%(n.replace("msmv_",""),v["p50_us"]) for n,v in d["kernels"].items()), flush=True)
PY
done
