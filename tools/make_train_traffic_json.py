#!/usr/bin/env python3
"""profiles/traffic_train.json from the PMC passes of tools/profile_train_pmc.sh: HBM-side bytes of ONE training step =
sum over every dispatch of the step of (FETCH_SIZE x 2 [gfx950: FETCH_SIZE reports half of the bytes of wide coalesced
reads, MI355X_MICROARCH.md HBM section] + WRITE_SIZE) KiB x 1024, with the largest contributors by kernel.
usage: make_train_traffic_json.py <prof dir> <git head> <steps incl. warmup> <out json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, head, nsteps, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("egnn::", "").split("(")[0]
    return name[-60:]


def total(sub, counter):
    acc = defaultdict(float)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


fetch, write = total("pmc_fetch", "FETCH_SIZE"), total("pmc_write", "WRITE_SIZE")
by_kernel = {k: (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / nsteps for k in set(fetch) | set(write)}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd import _lib  # noqa: E402  (no GPU call: only the source fingerprint)

doc = {"comment": __doc__.split("usage")[0].strip(), "git_head": head, "training_sources_sha256": _lib.training_sources_sha256(), "source": os.path.basename(out.rstrip("/")),
       "workload": "bench.py --mode train: 256 graphs x 64 atoms per rank, bf16, kept activations",
       "bytes_per_step": sum(by_kernel.values()),
       "largest": {k: v for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:14]}}
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(doc, indent=1))
