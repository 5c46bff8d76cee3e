#!/usr/bin/env python3
"""Turn the raw rocprofv3 output that tools/collect_profiles.sh left under gpurun_out/<tag>/
into the committed summaries: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and
profiles/traffic.json (L2 <-> fabric bytes per launch, keyed by workload | batch | points, which
bench.py reports as roofline.traffic).
    python tools/make_profile_summary.py TAG [WORKLOAD [BATCH [POINTS]]]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B read request:
it reads exactly half the known bytes for 16-byte-per-lane reads of 256-byte vectors (checked
with tools/calib_fetch.hip, whose counter value is stored alongside), so reads are doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Infinity Cache hits are included: the
counters sit on the L2's fabric side.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.buildinfo import csrc_hash  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "L-fwdbwd"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
points = sys.argv[4] if len(sys.argv) > 4 else "uniform"
out_tag = f"{tag}_{workload}"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    for k in ("msmv_fwd_c64", "msmv_fwd_sorted_c64", "msmv_bwd_owner_dfeat", "msmv_bwd_gather_c64", "msmv_bin_count",
              "msmv_bin_fill", "msmv_bin_one", "msmv_bwd_atomic", "msmv_fwd_generic", "msmv_project_pack_fwd",
              "msmv_project_pack_bwd", "msmv_relayout", "calib_gather"):
        if k in name:
            return k
    return None


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(stats)) if short(r["Name"])]
with open(os.path.join(dst, f"{out_tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{out_tag}_bench_under_rocprof.json"))

pmc = defaultdict(lambda: defaultdict(list))
template = {}
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            m = re.search(r"(msmv_\w+|calib_\w+)(<[^>]*>)?", r["Kernel_Name"])
            template[k] = m.group(0) if m else r["Kernel_Name"]
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}
calib = summary.get("calib_gather", {}).get("FETCH_SIZE")
known_calib_kib = ((1 << 21) * 256 + (1 << 21) * 4) / 1024.0
fetch_scale = round(known_calib_kib / calib, 3) if calib else 2.0
traffic = {}
for k, cs in summary.items():
    if k == "calib_gather" or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    rd, wr = cs["FETCH_SIZE"] * 1024 * fetch_scale, cs["WRITE_SIZE"] * 1024
    cs["hbm_side_read_bytes"] = rd
    cs["hbm_side_write_bytes"] = wr
    if "TCC_HIT_sum" in cs:
        cs["l2_hit_rate"] = cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
    # the bench's record of a kernel: bytes per launch, the kernel instance they were measured on
    # (bench.py's name of the kernel in this workload)
    bench_name = {"msmv_fwd_sorted_c64": "msmv_fwd_sorted", "msmv_project_pack_fwd": "msmv_project_pack"}.get(k, k)
    if k == "msmv_bwd_owner_dfeat" and (workload.startswith("decoder-") or workload == "train-M" or workload.startswith("chain-")):
        bench_name = "msmv_bwd_owner_dfeat(6 layers)"
    traffic[bench_name] = {
        "bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "template": template.get(k),
        **({"l2_hit_rate": cs["l2_hit_rate"]} if "l2_hit_rate" in cs else {})}
summary["_fetch_size_scale"] = fetch_scale
summary["_calibration"] = {"kernel": "calib_gather (tools/calib_fetch.hip)", "known_read_KiB": known_calib_kib,
                           "FETCH_SIZE": calib}
json.dump(summary, open(os.path.join(dst, f"{out_tag}_pmc.json"), "w"), indent=1, sort_keys=True)
tpath = os.path.join(dst, "traffic.json")
allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
allt.pop(workload, None)                 # (records of earlier rounds were keyed by the workload alone)
# the kernel sources the passes ran on: bench.py ignores the record once they change
traffic["_csrc_sha256"] = csrc_hash()
traffic["_profile"] = f"profiles/{out_tag}_pmc.json"
allt[f"{workload}|batch={batch}|points={points}"] = traffic
allt["_source"] = ("profiles/*_pmc.json (rocprofv3 --pmc passes of bench.py on exactly the keyed configuration, "
                   "see tools/collect_profiles.sh); bytes cross the L2 <-> fabric boundary, Infinity-Cache hits included")
json.dump(allt, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
for r in rows:
    print(short(r["Name"]).ljust(24), r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"])
