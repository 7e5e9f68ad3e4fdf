"""Where the set-up time of the fused quad plan goes (10 M plate unless --strip): python tools/exp/plan_setup_time.py"""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from fem_shell_amd import meshes, fused, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
nx, ny = (bench.STRIP_NX, bench.STRIP_NY) if "--strip" in sys.argv else (bench.FULL_NX, bench.FULL_NY)
lams = bench.c5_laminates()
mesh = meshes.jittered_plate(nx, ny, cell=1e-3, seed=11, n_sets=16, band=bench.BAND)
asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL,
                                        "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}}, cache_blocks=False)
torch.cuda.synchronize()
for rep in range(2):
    t = [time.perf_counter()]
    g = asm._groups[0]
    patch = fused.partition_nodes_device(mesh.node_count, g["conn_d"], 64); torch.cuda.synchronize(); t.append(time.perf_counter())
    p = asm._plan
    nbr_ptr, nbr = p.nbr_ptr.cpu().numpy(), p.nbr.cpu().numpy(); t.append(time.perf_counter())
    from types import SimpleNamespace
    host = SimpleNamespace(d=6, n_nodes=p.n_nodes, nbr_ptr=nbr_ptr, nbr=nbr)
    fp = fused.FusedQuadPlan(host, asm._conn, asm._coords, node_patch=patch); t.append(time.perf_counter())
    d = fp.to_device(asm.device); torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["partition (device) + D2H of the map", "node graph D2H", "native builder + export", "upload of the plan"]
    print(f"rep {rep}: " + ", ".join(f"{n} {b - a:.3f} s" for n, a, b in zip(names, t[:-1], t[1:])), f"total {t[-1] - t[0]:.3f} s; records {fp.recs.nbytes / 1e9:.2f} GB")
    del fp, d
# the device builder (fs_fused_plan_device_begin / _finish)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    patch = fused.partition_nodes_device(mesh.node_count, g["conn_d"], 64, as_host=False); torch.cuda.synchronize(); t1 = time.perf_counter()
    fpd = fused.FusedQuadPlanDevice(asm._plan, g["conn_d"], patch, 64); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"device rep {rep}: partition {t1 - t0:.3f} s, device builder {t2 - t1:.3f} s, total {t2 - t0:.3f} s; "
          f"{fpd.n_patches} patches, {fpd.n_recs * 64 / 1e9:.2f} GB of records")
    del fpd
