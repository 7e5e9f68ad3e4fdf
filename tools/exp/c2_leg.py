import sys, json, torch
sys.path.insert(0, '.')
import bench
r = bench.c2_leg(torch)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "ms" in kk} for k, v in r.items()}, indent=1))
