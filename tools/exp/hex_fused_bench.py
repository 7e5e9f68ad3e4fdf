"""Quick timing of the fused HEXA8 kernel on configs[4]'s block: python tools/exp/hex_fused_bench.py [nx ny nz]"""
import sys, time, json
sys.path.insert(0, '.')
import torch
import bench
out = bench.hex8_leg(torch, steps=5)
print(json.dumps({k: out[k] for k in ("value", "fused", "two_kernel_ms_per_step", "setup_s")}, indent=1))
for k, v in out["kernels"].items():
    print(k, "%.3f ms" % v["kernel_ms"])
