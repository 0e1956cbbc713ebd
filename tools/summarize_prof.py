#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel trace stats + PMC counters) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    for k in ("edge_kernel", "edge_x_m16_kernel", "edge_x3_kernel", "edge_dgrad_kernel", "node_post", "node_pre", "node_d2_kernel",
              "graph_scale_kernel", "sampler_step_kernel", "advance_t_kernel"):
        if k in name:
            return name[name.find(k):][:60]
    return name[-60:]


print("== kernel trace stats ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{short(r['Name']):60s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):>12.0f} "
              f"total_ns={float(r['TotalDurationNs']):>14.0f} pct={r['Percentage']}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    seen = set()
    for r in rows:
        k = short(r["Kernel_Name"])
        if k in seen:
            continue
        seen.add(k)
        print(f"{k:60s} vgpr={r.get('VGPR_Count')} accum={r.get('Accum_VGPR_Count')} sgpr={r.get('SGPR_Count')} "
              f"lds={r.get('LDS_Block_Size')} scratch={r.get('Scratch_Size')} grid={r.get('Grid_Size_X', r.get('Grid_Size'))} wg={r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}")
print("== PMC (mean per dispatch) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if not any(x in k for x in ("edge_kernel", "edge_x", "node_post", "node_pre")):
            continue
        print(f"[{os.path.basename(d)}] {k}")
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} mean={sum(v)/len(v):.4g}  n={len(v)}")
