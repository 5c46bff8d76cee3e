#!/usr/bin/env python3
"""Per-step HIP-event durations of the first N OPUS-L forward + backward steps after set-up
(no settle phase, no warm-up): shows what the first steps of a short run pay.
    python tools/ramp_probe.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
bench.load_product()
dev = torch.device("cuda:0")
feats, sets = bench.make_inputs(32, 4800, 2, 1234, dev)
step = bench.Step(feats, sets, True, "owner")
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    step.run(False)
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
print("step us:", " ".join(f"{x:.0f}" for x in t))
print("mean of steps 0-4 %.0f, 5-24 %.0f, 25-49 %.0f, 50-99 %.0f, last 20 %.0f" % (
    sum(t[:5]) / 5, sum(t[5:25]) / 20, sum(t[25:50]) / 25, sum(t[50:100]) / 50, sum(t[-20:]) / 20))
