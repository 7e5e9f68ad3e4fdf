#!/usr/bin/env python3
"""Per-kernel launch statistics from a rocprofv3 --kernel-trace CSV, launches of one kernel grouped by grid size (the
bench runs the same kernel on meshes of different sizes; an average over all of them says nothing).
usage: python tools/trace_summary.py <dir with *_kernel_trace.csv> > profiles/rNN_..._kernels.txt"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        import math
        # persistent kernels launch the same grid for every mesh size: split by duration, half-decade bins
        acc[(name, grid, round(2 * math.log10(max(us, 1e-3))))].append(us)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':70s} {'grid_x':>12s} {'launches':>8s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s}")
for (name, grid, _), v in rows[:60]:
    print(f"{name[:70]:70s} {grid:12d} {len(v):8d} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {sum(v) * 1e-3:10.2f}")
