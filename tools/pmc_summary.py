#!/usr/bin/env python3
"""Summarise gpurun_out/pmc/*/ counter_collection.csv -> per-kernel averages per launch; writes
profiles/pmc_summary.json (hbm bytes per launch with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md 'HBM')."""
import csv, glob, json, collections, sys, os, re
args = [a for a in sys.argv[1:] if not a.startswith("--")]
root = args[0] if args else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)", k)
        short = m.group(1) if m else None
        if short:
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    # FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B? rocprofv3 reports KB; guide: bytes = value*1024, FETCH x2 on gfx950
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_per_launch"] = d["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    out[k] = d
print(json.dumps(out, indent=1))
if "--write" in sys.argv:
    json.dump(out, open("profiles/pmc_summary.json", "w"), indent=1)
