#!/bin/bash
# Runs in the build container after `gpurun -- 'bash tools/collect_bench.sh r02; bash tools/collect_profiles.sh
# r02p; ... r02_s4d --workload sampling4d-L; ... r02_T --workload T-fwdbwd'`: turns what came back under
# gpurun_out/ into the committed files under profiles/.
set -e
cd "$(dirname "$0")/.."
rm -rf gpurun_out/r02/stats gpurun_out/r02/pmc_* gpurun_out/r02/calib*
cp -r gpurun_out/r02p/* gpurun_out/r02/ && rm -rf gpurun_out/r02p
python tools/make_profile_summary.py r02 L-fwdbwd 1 uniform > /dev/null
python tools/make_profile_summary.py r02_s4d sampling4d-L 1 uniform > /dev/null
python tools/make_profile_summary.py r02_T T-fwdbwd 1 uniform > /dev/null
cd profiles
for f in r02_s4d_sampling4d-L_*; do mv -f "$f" "${f/r02_s4d_sampling4d-L/r02_sampling4d-L}"; done
for f in r02_T_T-fwdbwd_*; do mv -f "$f" "${f/r02_T_T-fwdbwd/r02_T-fwdbwd}"; done
cd ..
cp gpurun_out/r02/bench/r02_*.json profiles/
