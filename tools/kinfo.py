#!/usr/bin/env python3
"""Register / LDS / occupancy figures of every kernel in an assembly listing made with
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o out.s file.hip"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
# the per-function comment block follows the function's body: "; Kernel Name" is not emitted, so
# walk the labels and the info blocks in order
for m in re.finditer(r"^(_Z\w+):[^\n]*\n.*?; TotalNumSgprs: (\d+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; LDSByteSize: (\d+).*?; Occupancy: (\d+)",
                     s, re.S | re.M):
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", m.group(1))
    if pat in name:
        print(f"{name[:48]:48s} sgpr {m.group(2):>3s} vgpr {m.group(3):>3s} scratch {m.group(4):>4s} lds {m.group(5):>6s} occupancy {m.group(6)}")
