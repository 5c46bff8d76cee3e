#!/bin/bash
# Runs ON THE GPU BOX: bench.py for every (library build, value of an environment switch) pair.
#   LIBS="a.so b.so" VAR=OPUS_MSMV_DFEAT_H16 VALS="0 1" WL=decoder-L PTS=uniform bash tools/ab_libs_env.sh
for lib in $LIBS; do
  for v in ${VALS:-0 1}; do
    env OPUS_MSMV_LIB=$PWD/$lib ${VAR}=$v timeout -k 10 200 python bench.py --workload ${WL:-L-fwdbwd} --points ${PTS:-uniform} --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null || { echo "FAILED $lib $v"; continue; }
    python - "$(basename $lib) $VAR=$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1].ljust(40), d["config"]["name"], d["config"]["points"], "step %.1f"%(d["ms_per_step"]*1e3), " ".join("%s=%.1f"%(n.replace("msmv_",""),v["p50_us"]) for n,v in d["kernels"].items() if "dfeat" in n), flush=True)
PY
  done
done
