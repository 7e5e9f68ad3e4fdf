import sys, json, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.newton_leg(torch), indent=1))
