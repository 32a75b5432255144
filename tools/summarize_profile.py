#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the small files kept under profiles/."""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
# what was profiled: tools/gpu.sh leaves the commit and the library stamp in profiles/BUILD_ID before the snapshot travels
# (.git stays behind); without it the library's own source digest
from pathlib import Path
_root = Path(__file__).resolve().parents[1]
_bid = _root / "profiles" / "BUILD_ID"
_stamp = _root / "primitive3d_amd" / "libp3dmc.so.stamp"
BUILD = _bid.read_text().strip() if _bid.exists() else ("lib=" + _stamp.read_text().strip()[:16] if _stamp.exists() else "unknown")
print("build:", BUILD)
ours = ("k_fused", "k_face_count_walk", "k_face_total", "k_faces", "k_stack_finish", "k_export_plane_records", "k_scan_blocks", "k_classify",
        "k_unit_counts", "k_unit_records", "k_emit_vertices", "k_fix_records")


def short(name):
    for o in ours:
        if o + "<" in name or o + "(" in name or name.endswith(o) or o + "I" in name or o + "E" in name:
            return o
    return None


for sub, label in (("stats", "headline c3"), ("stats_fresh", "c3, four distinct grids in turn (modes.fresh_grid)"), ("stats_c5", "c5 batch"),
                   ("stats_c2", "c2 bunny")):
    rows = []
    for f in glob.glob(d + f"/{sub}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    if not rows:
        continue
    print(f"== rocprofv3 --kernel-trace --stats, {label} (our kernels; ns)")
    keep = [r for r in rows if short(r["Name"])]
    for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"])):
        print(f'{short(r["Name"]):22s} calls={r["Calls"]:>4s} avg={float(r["AverageNs"]):10.0f} min={r["MinNs"]:>8s} max={r["MaxNs"]:>8s}')
    name = "kernel_stats_ours.csv" if sub == "stats" else f"kernel_{sub}_ours.csv"
    with open(d + "/" + name, "w") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in keep:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])


def counter(sub, cname):
    acc = defaultdict(list)
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s and r["Counter_Name"] == cname:
                acc[s].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = {}
fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
cf, cw = counter("cal_fetch", "FETCH_SIZE"), counter("cal_write", "WRITE_SIZE")
print("== PMC (raw counter units = KiB per the rocprofv3 definition), averaged per launch")
print("FETCH_SIZE", {k: round(v) for k, v in fetch.items()})
print("WRITE_SIZE", {k: round(v) for k, v in write.items()})
print("calibration FETCH_SIZE", {k: round(v) for k, v in cf.items()}, "WRITE_SIZE", {k: round(v) for k, v in cw.items()})
known_read = 512 ** 3 * 4  # k_classify reads the field exactly once
if "k_classify" in cf and cf["k_classify"] > 0:
    corr = known_read / (cf["k_classify"] * 1024)
    print(f"read correction factor for this access shape (dword/lane streaming): {corr:.3f}")
    if "k_fused" in fetch:
        rd = fetch["k_fused"] * 1024 * corr
        wr = write.get("k_fused", 0) * 1024
        out = {"k_fused_hbm_bytes_per_launch": round(rd + wr), "k_fused_read_bytes": round(rd), "k_fused_write_bytes": round(wr),
               "fetch_correction": round(corr, 4), "calibrated_on": "k_classify reading 512^3 fp32 once (known 536870912 B)",
               "build": BUILD}
        print(out)
json.dump(out, open(d + "/traffic.json", "w"), indent=1)
