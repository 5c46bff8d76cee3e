#!/bin/bash
# Builds ablation / experiment variants of the library: abl/build.sh NAME "-DFLAG ..." [NAME2 "..."]...
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-value -w"
SRCS="msmv_api.hip msmv_forward.hip msmv_backward.hip msmv_project.hip msmv_relayout.hip"
while [ $# -gt 0 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -shared -o abl/lib_$name.so $SRCS &
done
wait
ls -la abl/*.so
