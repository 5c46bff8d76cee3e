#!/bin/bash
# Runs in the build container after tools/collect_profiles.sh has been run on the GPU box for the six
# profiled workloads: turns gpurun_out/<tag>_*/ into profiles/<tag>_*_{kernel_stats.csv,pmc.json,...}
# and profiles/traffic.json (stamped with the kernel sources' hash).  The bench lines (tools/collect_bench.sh)
# are collected AFTER this, so that they carry the traffic measured on their own sources.
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}
python tools/make_profile_summary.py ${TAG}_L L-fwdbwd 1 uniform > /dev/null
cd profiles; for f in ${TAG}_L_L-fwdbwd_*; do [ -e "$f" ] && mv -f "$f" "${f/${TAG}_L_L-fwdbwd/${TAG}_L-fwdbwd}"; done; cd ..
sed -i "s#profiles/${TAG}_L_L-fwdbwd_pmc.json#profiles/${TAG}_L-fwdbwd_pmc.json#" profiles/traffic.json
for w in T-fwdbwd sampling4d-L decoder-L train-M chain-L; do
    python tools/make_profile_summary.py ${TAG}_$w $w 1 uniform > /dev/null
    cd profiles; for f in ${TAG}_${w}_${w}_*; do [ -e "$f" ] && mv -f "$f" "${f/${TAG}_${w}_${w}/${TAG}_${w}}"; done; cd ..
    sed -i "s#profiles/${TAG}_${w}_${w}_pmc.json#profiles/${TAG}_${w}_pmc.json#" profiles/traffic.json
done
