#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_run.sh) -> per-kernel averages per launch; writes
profiles/pmc_summary.json.  HBM bytes per launch with the gfx950 corrections of MI355X_MICROARCH.md 'HBM'
(FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x 2).

  gpurun_out/pmc/strip/<pass>/   bench.py --strip-only --no-legs: the fused kernel on the 1.25 M strip, nothing else
  gpurun_out/pmc/legs/<pass>/    bench.py --strip-only (with the legs): every other kernel; a kernel launched on several
                                 mesh sizes is summarised over its LARGEST grid only (the 1.25 M / 1 M element launches)
usage: python tools/pmc_summary.py [root] [--write]"""
import collections
import csv
import glob
import json
import os
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
root = args[0] if args else "gpurun_out/pmc"


def short_name(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+?)(<[^(]*>)?\(", k)
    name = m.group(1) if m else k
    return name.split("::")[-1]


def collect(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for f in glob.glob(os.path.join(root, sub, "*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short_name(r["Kernel_Name"])][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def summarise(grids):
    g = max(grids)
    d = {c: sum(v) / len(v) for c, v in grids[g].items()}
    d["grid_size"] = g
    d["launches_per_pass"] = min(len(v) for v in grids[g].values())
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_per_launch"] = d["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    return d


out = {}
legs, strip = collect("legs"), collect("strip")
for k, grids in legs.items():
    if k.endswith("_kernel") and k != "fused_quad_k_kernel":
        out[k] = summarise(grids)
if "fused_quad_k_kernel" in strip:
    out["fused_quad_k_kernel"] = summarise(strip["fused_quad_k_kernel"])
    out["fused_quad_k_kernel"]["elements"] = 1250000
    out["fused_quad_k_kernel"]["workload"] = "400x3125 strip of the C5 plate (bench.py --strip-only --no-legs)"
print(json.dumps(out, indent=1))
if "--write" in sys.argv:
    json.dump(out, open("profiles/pmc_summary.json", "w"), indent=1)
