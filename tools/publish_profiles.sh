#!/bin/bash
# Runs in the build container after
#   gpurun -- 'bash tools/collect_bench.sh r03 && bash tools/collect_profiles.sh r03_L &&
#              for w in T-fwdbwd sampling4d-L decoder-L train-M chain-L; do bash tools/collect_profiles.sh r03_$w --workload $w || exit 1; done'
# turns what came back under gpurun_out/ into the committed files under profiles/ (kernel stats, PMC
# summaries, traffic.json keyed by workload | batch | points and stamped with the kernel sources' hash).
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
python tools/make_profile_summary.py ${TAG}_L L-fwdbwd 1 uniform > /dev/null
cd profiles; for f in ${TAG}_L_L-fwdbwd_*; do [ -e "$f" ] && mv -f "$f" "${f/${TAG}_L_L-fwdbwd/${TAG}_L-fwdbwd}"; done; cd ..
sed -i "s#profiles/${TAG}_L_L-fwdbwd_pmc.json#profiles/${TAG}_L-fwdbwd_pmc.json#" profiles/traffic.json
for w in T-fwdbwd sampling4d-L decoder-L train-M chain-L; do
    pts=uniform      # (the key is what the command line said; caller-side workloads project camera points either way)
    python tools/make_profile_summary.py ${TAG}_$w $w 1 $pts > /dev/null
    cd profiles; for f in ${TAG}_${w}_${w}_*; do [ -e "$f" ] && mv -f "$f" "${f/${TAG}_${w}_${w}/${TAG}_${w}}"; done; cd ..
    sed -i "s#profiles/${TAG}_${w}_${w}_pmc.json#profiles/${TAG}_${w}_pmc.json#" profiles/traffic.json
done
cp gpurun_out/${TAG}/bench/${TAG}_*.json profiles/
