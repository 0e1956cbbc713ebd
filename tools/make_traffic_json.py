#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile.sh: HBM-side bytes per fused edge pass of one EGCL layer =
sum over the edge kernels of a layer of (FETCH_SIZE x 2 [gfx950: FETCH_SIZE reports half of the bytes of wide coalesced
reads, MI355X_MICROARCH.md HBM section] + WRITE_SIZE) KiB x 1024, mean per dispatch; plus MFMA busy fraction and the held
clock (GRBM_GUI_ACTIVE / 8 / kernel time).   usage: make_traffic_json.py <prof dir> <git head> <out json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, head, dst = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    i = name.find("edge_")
    return name[i:] if i >= 0 else name


def pmc(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("edge_")}


avg_ns = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("edge_"):
            avg_ns[k] = float(r["AverageNs"])
fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
kernels = sorted(avg_ns)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd import _lib  # noqa: E402  (no GPU call: only the source fingerprint and the library path)
import hashlib  # noqa: E402
doc = {"comment": __doc__.split("usage")[0].strip(), "git_head": head, "source": os.path.basename(out.rstrip("/")),
       "edge_kernel_sources_sha256": _lib.edge_kernel_sources_sha256(),
       "library_sha256_on_the_profiled_box": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest(),
       "workload": "256 graphs x 64 atoms, bf16, default path", "kernels": {}}
total = 0.0
for k in kernels:
    f = fetch.get(k, {}).get("FETCH_SIZE")
    w = write.get(k, {}).get("WRITE_SIZE")
    row = {"avg_launch_ms_under_rocprofv3": avg_ns[k] / 1e6, "fetch_size_kib": f, "write_size_kib": w}
    if f is not None and w is not None:
        row["hbm_bytes"] = (2 * f + w) * 1024
        total += row["hbm_bytes"]
    g = fetch.get(k, {}).get("GRBM_GUI_ACTIVE")
    if g:
        row["held_clock_ghz"] = g / 8 / avg_ns[k]
    m, wc = sq.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES"), sq.get(k, {}).get("SQ_BUSY_CYCLES")
    if m:
        row["sq_valu_mfma_busy_cycles"] = m
    doc["kernels"][k] = row
doc["bytes_per_launch"] = total
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(doc, indent=1))
