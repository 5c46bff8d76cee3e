set -e
for lib in opus_amd/csrc/abl/lib_s100000.so opus_amd/csrc/abl/lib_s64.so; do
  for pts in uniform cameras; do
    OPUS_MSMV_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python bench.py --workload ${WL:-L-fwdbwd} --steps 20 --warmup 5 --no-cpu-baseline --points $pts > gpurun_out/abl_tmp.json 2>/dev/null
    python - "$lib" $pts <<'PY'
import json,sys
d=json.load(open("gpurun_out/abl_tmp.json"))
k=d["kernels"]
print(sys.argv[1] or "base", sys.argv[2], "step %.1f"%(d["ms_per_step"]*1e3), " ".join("%s=%.1f"%(n.replace("msmv_",""),v["p50_us"]) for n,v in k.items()), flush=True)
PY
  done
done
