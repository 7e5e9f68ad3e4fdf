#!/usr/bin/env python3
"""bench.py -- MITC4+ element stiffness matrices assembled into CSR per second (fp64).

One "step" = one pass of the hot path over the rank's shard: ONE launch of the fused evaluate-and-assemble kernel
(fs_fused_assemble_quad: element K of every element and the element-order CSR sums, no element matrices in memory;
+ interface-row exchange when N > 1).  Inputs (coordinates, property table, patch plan, CSR pattern) are resident in
HBM before the timed region; the symbolic phase (pattern + plans) is precomputed, as north_star specifies.

Workload (BASELINE.json configs[4], the configuration the metric is quoted on): the synthetic 3200 x 3125 = 10 M element
MITC4+ laminate plate (16 laminates in x-bands of 25 columns cycling over the plate, jittered nodes).
  N = 1   the whole 10 M plate on one GPU is `value`; the 400 x 3125 strip one rank owns at N = 8 is timed in the same
          run and reported as `one_rank_share_at_n8`.
  N > 1   the SAME plate cut into N x-strips of 3200/N x 3125 elements (STRONG scaling: total work fixed, N = 8 is
          1.25 M elements per rank), interface rows exchanged with the two neighbours over RCCL.
configs[0..3] and the HEXA8 block of configs[4] are timed in their own legs outside `value`; tests/ holds their parity.

Contract: python bench.py --gpus N --steps K --warmup W  (N > 1: launched by torch.distributed.run,
one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.  W untimed warm-up steps, then exactly K timed steps between
barrier + synchronize; before the warm-up steps the same step runs untimed for about 100 ms (`config.prewarm_steps`,
see plate_run: the GPU's clocks settle over ~35 ms of load, which is three steps of the plate and thirty of a strip).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STRIP_NX, STRIP_NY = 400, 3125
FULL_NX, FULL_NY = 3200, 3125
BAND = 25                         # columns per laminate band of the plate (global column index; 16 laminates cycle)
# bench/FLOPS.md (tools/count_flops.py: fp64 instruction count of the shipped device functions, FMA = 2):
FLOP_PER_ELEM = 18779.0           # algorithmic, geometry once per element
FLOP_PER_ELEM_COUPLED = 21783.0   # membrane-bending coupling active (asymmetric laminates)
BYTES_PER_ELEM = 2708.0           # SURVEY.md 8(d) fused path: 96 B coordinates + 4 B property id + 4 B status
                                  # + 324 amortised CSR values x 8 B (+ 12 B of node / pattern data it rounds in)
PEAK_FP64_TFLOPS = 78.6           # MI355X fp64 vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec


def log(msg):
    """Progress on stderr (the one JSON line goes to stdout)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def avg_ms(torch, fn, steps):
    """Average GPU time of fn() over `steps` back-to-back calls, HIP events on the launch stream (torch's current
    stream is the stream every fs_* call is given)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    # steady clocks (see plate_run): the legs start after host-side set-up, i.e. after an idle gap -- run untimed for ~60 ms
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) < 0.06 and n < 200 and not os.environ.get("FEMSHELL_BENCH_NO_PREWARM"):
        fn()
        torch.cuda.synchronize()
        n += 1
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def pmc_traffic(full):
    """HBM bytes per launch of the fused kernel from the PMC counters, measured NOW (not quoted from profiles/): two
    child runs of this file under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE each in a pass of its own, as
    MI355X_MICROARCH.md (HBM, rocprofv3) prescribes -- on the same workload as `value` (children run `--pmc-child`:
    that workload only, 2 timed launches).  gfx950 corrections of the guide: both counters in KiB, FETCH_SIZE tallies the
    128-byte requests of wide coalesced reads at 64 B (x 2).  Returns (bytes per launch or None, details)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, {"error": "rocprofv3 not on PATH"}
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="femshell_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--steps", "2", "--warmup", "1"] + ([] if full else ["--strip-only"])
        env = dict(os.environ, TMPDIR="/tmp", FEMSHELL_BENCH_NO_PREWARM="1")      # (counters, not times)
        try:
            # own process group: a pass that hangs is killed with everything it started (the profiled python included)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=180)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                raise
            if rc != 0:
                raise RuntimeError(f"rocprofv3 exit code {rc}")
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "fused_quad_k_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, {"error": f"no {counter} rows for the fused kernel"}
            vals.sort()
            got[counter] = vals[len(vals) // 2]          # median over the child's launches (same workload each)
            got[counter + "_launches"] = len(vals)
        except Exception as ex:  # noqa: BLE001  (no profiler, no counters: the bench line says so instead of failing)
            return None, {"error": f"{counter} pass failed: {type(ex).__name__}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = got["FETCH_SIZE"] * 1024 * 2, got["WRITE_SIZE"] * 1024
    return rd + wr, {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "launches_sampled": got["FETCH_SIZE_launches"],
                     "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child runs of this bench on the same workload, "
                            "started by this run; KiB units, FETCH_SIZE x 2 (gfx950)"}


def c5_laminates():
    """16 laminates of the synthetic plate: UD, cross-ply, biax, QI, sandwich-like thick stacks,
    ~25 % asymmetric (membrane-bending coupling active)."""
    import fem_shell_amd as fsa
    m = fsa.OrthotropicMaterial("T300/5208", (181e9, 10.3e9, 10.3e9), (7.17e9, 3.0e9, 7.17e9),
                                (0.28, 0.28, 0.28), 1600.0)
    core = fsa.OrthotropicMaterial("core", (0.13e9, 0.13e9, 0.13e9), (0.05e9, 0.05e9, 0.05e9),
                                   (0.3, 0.3, 0.3), 100.0)
    t = 1.25e-6  # ply thickness for a 1e-3 cell plate (t/L ~ 1e-2 total)
    stacks = [
        [0] * 4, [0, 90, 90, 0], [45, -45, -45, 45], [0, 45, -45, 90, 90, -45, 45, 0],
        [0, 90], [0, 45, 90], [0, 90, 0, 90], [30, -30, 30],
        [0, 90] * 4 + [90, 0] * 4, [0, 45, -45, 90] * 2 + [90, -45, 45, 0] * 2,
        [60, -60, 0, 0, -60, 60], [0, 0, 90, 90, 0, 0], [15, -15, -15, 15], [90] * 6,
        [0, 30, 60, 90, 60, 30, 0], [45, -45] * 5 + [-45, 45] * 5,
    ]
    lams = []
    for i, ang in enumerate(stacks):
        plies = [fsa.Ply(m, t, a) for a in ang]
        if i in (9, 15):  # sandwich-like: thick soft core in the middle
            h = len(plies) // 2
            plies = plies[:h] + [fsa.Ply(core, 8 * t, 0.0)] + plies[h:]
        lams.append(fsa.CompositeShellProperty(fsa.Laminate(plies)))
    return lams


def host_threads():
    """Threads for the all-core figure: the CPUs this process may really use.  On the GPU box os.cpu_count() shows the
    whole machine while the job owns a 16-CPU share (the harness exports OMP_NUM_THREADS = 16 for that reason)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return n


def cpu_baseline():
    """Oracle (CPU restatement of the reference algorithm, oracle/) on bounded samples of the same workload.  `value` is
    the all-core figure: element K on every usable host thread + row-parallel element-order CSR summation
    (fso_csr_numeric_rows, bit-identical to the serial loop) on the same threads.  Also reported: one core, and the
    route of the reference's own benchmark (tests/benchmarks/test_ko2017_performance.py:365-398): element matrices ->
    COO triplets -> scipy coo_matrix.tocsr().  Reported baselines only."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from oracle import oracle as O
    from scipy.sparse import coo_matrix
    nt = host_threads()
    table = fsa.build_prop_table(c5_laminates())

    def sample(nx, ny):
        mesh = meshes.jittered_plate(nx, ny, n_sets=16)
        dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
        return mesh, mesh.coords[mesh.conn], mesh.set_id.astype(np.int32), dofs

    def run(nx, ny, threads):
        mesh, xyz, pid, dofs = sample(nx, ny)
        indptr, indices, _ = O.csr_assemble(dofs, mesh.node_count * 6, None)      # symbolic: untimed
        O.mitc4_K_batch(xyz[:1000], pid[:1000], table, nthreads=threads)          # warm the library
        t0 = time.perf_counter()
        Ke, _ = O.mitc4_K_batch(xyz, pid, table, nthreads=threads)
        t1 = time.perf_counter()
        O.csr_numeric_rows(dofs, mesh.node_count * 6, Ke, indptr, indices, nthreads=threads)
        t2 = time.perf_counter()
        E = xyz.shape[0]
        return E, E / (t2 - t0), E / (t1 - t0), E / (t2 - t1)

    nx, ny = 400, 1250          # 500 000 elements, 2.3 GB of dense element matrices
    E, v_all, k_all, a_all = run(nx, ny, nt)
    E1, v_one, k_one, a_one = run(400, 100, 1)
    # the reference benchmark's route, vectorised: all-core oracle K, NumPy triplets, SciPy COO -> CSR
    mesh, xyz, pid, dofs = sample(400, 250)
    t0 = time.perf_counter()
    Ke, _ = O.mitc4_K_batch(xyz, pid, table, nthreads=nt)
    rows = np.repeat(dofs, 24, axis=1).reshape(-1)
    cols = np.tile(dofs, (1, 24)).reshape(-1)
    n = mesh.node_count * 6
    Kc = coo_matrix((Ke.reshape(-1), (rows, cols)), shape=(n, n)).tocsr()
    v_coo = xyz.shape[0] / (time.perf_counter() - t0)
    return {"value": v_all, "unit": "elements/s", "cores": nt, "kind": "port",
            "sample": f"{nx}x{ny}={E} element piece of the same jittered laminate plate; oracle C element K on {nt} "
                      f"OpenMP threads ({k_all:.3e} el/s) + row-parallel element-order CSR summation on {nt} threads "
                      f"({a_all:.3e} el/s)",
            "host_cpu_count": os.cpu_count(),
            "single_core": {"value": v_one, "unit": "elements/s", "cores": 1, "element_K_el_per_s": k_one,
                            "csr_summation_el_per_s": a_one, "sample": f"400x100={E1} elements"},
            "scipy_coo_route": {"value": v_coo, "unit": "elements/s", "cores": nt, "nnz": int(Kc.nnz),
                                "sample": f"400x250={xyz.shape[0]} elements; oracle element K on {nt} threads, NumPy "
                                          "triplets, scipy.sparse.coo_matrix(...).tocsr() (single-threaded in SciPy)"}}


def hex8_leg(torch, steps=5):
    """configs[4] also names a HEXA8 block of 1 048 576 bricks (128 x 128 x 64), here whole on one GPU.  The reference
    cannot put shells and solids into one MeshAssembler (element_family is a single value, elements/elements.py:236-256),
    so it is a second, SOLID-family assembler; timed on its own, outside `value`."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    nx, ny, nz = 128, 128, 64
    ids = np.arange((nx + 1) * (ny + 1) * (nz + 1), dtype=np.int64).reshape(nz + 1, ny + 1, nx + 1)
    Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
    coords = np.stack([X, Y, Z], -1).reshape(-1, 3) * 1e-3
    coords += (meshes.hash_uniform(np.arange(coords.size), 5).reshape(-1, 3) - 0.5) * 1e-4
    n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
    conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)],
                    -1).reshape(-1, 8)
    mat = fsa.OrthotropicMaterial("T300/5208", (181e9, 10.3e9, 10.3e9), (7.17e9, 3.0e9, 7.17e9),
                                  (0.28, 0.28, 0.28), 1600.0)
    t0 = time.perf_counter()
    asm = MeshAssembler(meshes.ArrayMesh(coords, conn),
                        {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
    K = asm.assemble_stiffness_matrix(recompute=True)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    k_ms = avg_ms(torch, lambda: asm.refresh_element_matrices(want_K=True, want_M=False, check=False), steps)
    a_ms = avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=False, fused=False), steps)
    E = conn.shape[0]
    # the fused evaluate-and-assemble kernel (fs_fused_assemble_hex8): the default of assemble_stiffness_matrix(recompute=True)
    fused = None
    if asm.fused_ready():
        t1 = time.perf_counter()
        f_ms = avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=True), steps)
        Kf = asm.assemble_stiffness_matrix(recompute=True)
        K2 = asm.assemble_stiffness_matrix(recompute=True, fused=False)
        dev_rel = float((Kf.data - K2.data).abs().max() / K2.data.abs().max())
        fb = 192.0 + 8.0 * K.nnz / E          # SURVEY.md 8(d) form: coordinates in, CSR values out
        gbs = E * fb / (f_ms * 1e-3) / 1e9
        fused = {"ms_per_step": f_ms, "elements_per_s": E / (f_ms * 1e-3), "max_rel_dev_from_two_kernel_route": dev_rel,
                 "plan": asm.fused_plan().stats(),
                 "roofline": {"kernel": "fused_hex8_kernel", "kernel_ms": f_ms, "bound": "hbm", "achieved": gbs,
                              "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "bytes_per_element": fb,
                              "note": "algorithmic bytes: 192 B coordinates + 8 nnz / E CSR values; the kernel contracts over the "
                                      "Gauss points first (K = C : sum_q g g^T): about 15 kflop per element executed incl. the "
                                      "halo, against 47 kflop of the B^T C B form (SURVEY.md 8d)"}}
        del Kf, K2
        # the consistent mass matrix through the same plan (fs_fused_assemble_hex8_mass) against its two-kernel route
        fused["mass_ms_per_step"] = avg_ms(torch, lambda: asm.assemble_mass_matrix(recompute=True), steps)
        fused["mass_two_kernel_ms_per_step"] = avg_ms(torch, lambda: asm.assemble_mass_matrix(recompute=True, fused=False), steps)
        asm._M_blk = None
    flop = 47000.0                    # SURVEY.md 8(d) estimate for the 2x2x2 HEXA8 stiffness (not re-derived)
    # element kernel: 192 B coordinates in, 36 upper 3x3 node-pair blocks out; assembly: those blocks in, CSR values out
    a_bytes = 36 * 72 + 8.0 * K.nnz / E
    return {"workload": f"{nx}x{ny}x{nz}={E} jittered HEXA8, orthotropic C, SOLID-family assembler; value = fused "
                        "evaluate-and-assemble kernel (one launch per step), two_kernel_* = element kernel + assembly kernel",
            "value": (E / (fused["ms_per_step"] * 1e-3)) if fused else E / ((k_ms + a_ms) * 1e-3), "unit": "elements/s",
            "fused": fused, "two_kernel_ms_per_step": k_ms + a_ms, "two_kernel_elements_per_s": E / ((k_ms + a_ms) * 1e-3),
            "nnz": int(K.nnz), "setup_s": setup,
            "kernels": {
                "solid_kernel<8>": {"kernel_ms": k_ms, "bound": "valu_fp64", "achieved": E * flop / (k_ms * 1e-3) / 1e12,
                                    "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                                    "frac": E * flop / (k_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS, "flop_per_element": flop,
                                    "note": "priced with the 47 kflop of the B^T C B form (SURVEY.md 8d); the kernel "
                                            "contracts over the Gauss points first and executes about 13 kflop"},
                "csr_numeric_desc3_lds_kernel": {"kernel_ms": a_ms, "bound": "hbm", "achieved": E * a_bytes / (a_ms * 1e-3) / 1e9,
                                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                 "frac": E * a_bytes / (a_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                 "bytes_per_element": a_bytes}}}


def c2_leg(torch, steps=20):
    """configs[1]: pinched hemisphere with an 18 degree cut-out, 256 x 256, MITC4+ (65 536 quads: the fused quad kernel)
    and MITC3+ (131 072 triangles: the fused MITC3+ kernel, element kernel + assembly kernel beside it) on one GPU
    (SURVEY.md 8d C2)."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    mat = fsa.IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0)
    out = {}
    # the third entry is not configs[1]: the same element on a mesh large enough to fill the chip (1 024 000 triangles)
    for tag, tri in (("mitc4_256x256", False), ("mitc3_256x256", True), ("mitc3_cylinder_1000x512", True)):
        mesh = (meshes.cylindrical_patch(25.0, 25.0, 40.0, 1000, 512, triangular=True) if tag.startswith("mitc3_cyl")
                else meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 256, triangular=tri))
        asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.04}})
        K = asm.assemble_stiffness_matrix(recompute=True)
        E = mesh.element_count
        ms = avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=True), steps)
        out[tag] = {"elements": E, "nnz": int(K.nnz), "ms_per_step": ms, "elements_per_s": E / (ms * 1e-3)}
        if tri:
            # the step is the fused MITC3+ kernel (one launch); the two-kernel route it replaced, for comparison
            fb = 72 + 4 + 4 + 8 * int(K.nnz) / E          # coordinates, property id, status in; CSR values out
            out[tag]["fused"] = asm.fused_ready()
            out[tag]["fused_plan"] = {k: asm.fused_plan().stats()[k] for k in ("patches", "stage1_recompute", "step_fill")}
            out[tag]["roofline"] = {"kernel": "fused_tri_kernel", "kernel_ms": ms, "bound": "hbm",
                                    "achieved": E * fb / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": E * fb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_element": fb,
                                    "note": ("962 patches on 256 CUs: launch latency and the tail weigh in at this size"
                                             if E < 500000 else "7 436 patches: 29 per CU")}
            out[tag]["two_kernel_ms_per_step"] = avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=True, fused=False), steps)
            # the consistent mass matrix through the same plan (fs_fused_assemble_tri_mass) and through its two kernels
            out[tag]["mass_ms_per_step"] = avg_ms(torch, lambda: asm.assemble_mass_matrix(recompute=True), steps)
            out[tag]["mass_two_kernel_ms_per_step"] = avg_ms(torch, lambda: asm.assemble_mass_matrix(recompute=True, fused=False), steps)
            k_ms = avg_ms(torch, lambda: asm.refresh_element_matrices(want_K=True, want_M=False, check=False), steps)
            kb = 72 + 4 + 4 + 6 * 288          # 3 nodes in, 6 upper node-pair blocks out
            out[tag]["kernels"] = {"mitc3_k_lds_kernel": {
                "kernel_ms": k_ms, "bound": "hbm", "achieved": E * kb / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": E * kb / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_element": kb,
                "note": "131 072 elements = 2 048 waves on 1 024 SIMDs: launch latency and the tail weigh in at this size"}}
        else:
            out[tag]["fused_plan"] = {k: asm.fused_plan().stats()[k] for k in ("patches", "stage1_recompute", "stage2_lane_fill")}
        del asm
    return out


def measured_peaks(torch):
    """The box's own ceilings, measured in this run: fp64 FMA issue rate with one and with four waves
    per SIMD (8 independent chains per lane), HBM write-only (memset), read+write (copy) and read-only (sum)."""
    from fem_shell_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    out = torch.empty(8192 * 64, dtype=torch.float64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = {}
    iters = 20000
    for tag, waves in (("fp64_fma_1_wave_per_simd_TFLOPs", 1024), ("fp64_fma_4_waves_per_simd_TFLOPs", 4096)):
        _lib.check(lib.fs_bench_fp64_fma(waves, 10, _lib.ptr(out), _lib.stream_ptr()), "fs_bench_fp64_fma")
        torch.cuda.synchronize()
        ev[0].record()
        _lib.check(lib.fs_bench_fp64_fma(waves, iters, _lib.ptr(out), _lib.stream_ptr()), "fs_bench_fp64_fma")
        ev[1].record()
        torch.cuda.synchronize()
        res[tag] = 2.0 * 64 * 64 * iters * waves / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12
    n = 1 << 29                                    # 4 GiB of doubles: far beyond the 256 MB Infinity Cache
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) * 1e-3 / reps

    res["hbm_write_GBs"] = 8.0 * n / timed(lambda: a.zero_()) / 1e9
    res["hbm_copy_GBs"] = 16.0 * n / timed(lambda: b.copy_(a)) / 1e9
    res["hbm_read_GBs"] = 8.0 * n / timed(lambda: a.sum()) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return res


def c1_leg(torch, steps=50):
    """configs[0]: Scordelis-Lo roof, MITC4+ 32 x 32 (1 024 elements, 6 534 dofs, nnz 338 724): the reference's own
    CPU-runnable case.  GPU: element kernel + assembly per step (launch-bound at this size), device Dirichlet
    reduction, SciPy solve on the host as in the reference's benchmark; CPU: the oracle on one core."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.bc import BoundaryConditionManager, DirichletCondition
    from fem_shell_amd.elements import ElementFamily
    from oracle import oracle as O
    from scipy.sparse.linalg import spsolve
    R, L, ang, t, rho, n = 25.0, 25.0, 40.0, 0.25, 360.0, 32
    mesh = meshes.cylindrical_patch(R, L, ang, n, n)
    mat = fsa.IsotropicMaterial("Ko2017_Scordelis", 4.32e8, 0.0, rho)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": t}})
    for _ in range(5):
        K = asm.assemble_stiffness_matrix(recompute=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        K = asm.assemble_stiffness_matrix(recompute=True)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / steps * 1e3
    # linear static solve (gravity lumped to the nodes, symmetry + diaphragm constraints of the benchmark)
    F = np.zeros(mesh.node_count * 6)
    for row in mesh.conn:
        c = mesh.coords[row]
        area = 0.5 * np.linalg.norm(np.cross(c[1] - c[0], c[2] - c[0])) + 0.5 * np.linalg.norm(np.cross(c[2] - c[0], c[3] - c[0]))
        F[row * 6] -= area * t * rho / 4
    fixed = []
    for i, (x, y, z) in enumerate(mesh.coords):
        if abs(z) < 1e-6:
            fixed += [i * 6 + 2, i * 6 + 3, i * 6 + 4]
        if abs(z - L) < 1e-6:
            fixed += [i * 6 + 0, i * 6 + 1, i * 6 + 5]
        if abs(y) < 1e-6 and x > 0.0:
            fixed += [i * 6 + 1, i * 6 + 3, i * 6 + 5]
    bcm = BoundaryConditionManager(K, F, dof_per_node=6)
    bcm.apply_dirichlet([DirichletCondition(sorted(set(fixed)), 0.0)])
    K_red, F_red, _ = bcm.reduced_system
    u = bcm.expand_solution(spsolve(K_red.to_scipy().tocsc(), F_red.cpu().numpy())).cpu().numpy()
    node = int(np.argmin(np.abs(mesh.coords - np.array([R * np.cos(np.radians(ang)), R * np.sin(np.radians(ang)), 0.0])).sum(axis=1)))
    # CPU: oracle element K + element-order CSR summation, one core
    xyz = mesh.coords[mesh.conn]
    dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
    O.csr_assemble(dofs, mesh.node_count * 6, O.mitc4_K_batch(xyz, asm._prop_id, asm._prop_table)[0])
    t0 = time.perf_counter()
    for _ in range(5):
        Ke, _ = O.mitc4_K_batch(xyz, asm._prop_id, asm._prop_table)
        O.csr_assemble(dofs, mesh.node_count * 6, Ke)
    cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    return {"elements": mesh.element_count, "nnz": int(K.nnz), "gpu_ms_per_assembly": gpu_ms,
            "cpu_oracle_ms_per_assembly_1_core": cpu_ms, "normalised_displacement": abs(u[node * 6]) / 3.0240e-1,
            "paper_value_n16": 0.9973, "note": "normalised displacement on this 32 x 32 mesh; the reference's benchmark "
            "pins the 16 x 16 value (0.9973, 5 % band); oracle time includes its symbolic pass; the reference's "
            "per-object Python path runs at about 580 elements/s/core (BASELINE.md)"}


def c3_leg(torch, steps=10):
    """configs[2]: blade-like laminate shell (tapered closed tube, 512 x 128 quads, 192 property sets drawn from UD /
    biax / triax / sandwich layups, ~10 % asymmetric -- the real blade cannot be regenerated here) with
    its rotor loads: per step K (element kernel + assembly), M (element kernel + assembly), row-sum lumping and the
    centrifugal nodal forces at 0.79 rad/s about z.  Timed on its own, outside `value`."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    from fem_shell_amd.inertial import InertialForcesCalculator
    mesh = meshes.tapered_tube(128, 512)
    rng = np.random.default_rng(7)
    m = fsa.OrthotropicMaterial("T300/5208", (181e9, 10.3e9, 10.3e9), (7.17e9, 3.0e9, 7.17e9), (0.28, 0.28, 0.28), 1600.0)
    core = fsa.OrthotropicMaterial("core", (0.13e9,) * 3, (0.05e9,) * 3, (0.3,) * 3, 100.0)
    props = {}
    for k, name in enumerate(mesh.set_names):
        kind = rng.integers(0, 4)
        if kind == 3:
            lam = fsa.Laminate([fsa.Ply(m, 1e-3, 0.0), fsa.Ply(m, 1e-3, 45.0), fsa.Ply(core, 25e-3, 0.0),
                                fsa.Ply(m, 1e-3, 45.0), fsa.Ply(m, 1e-3, 0.0)])
        else:
            ang = ([0.0] * int(rng.integers(4, 41)), [45.0, -45.0] * int(rng.integers(2, 21)),
                   [0.0, 45.0, -45.0] * int(rng.integers(2, 14)))[kind]
            lam = fsa.create_laminate_from_angles(m, 0.5e-3, ang + ([90.0] if k % 10 == 0 else []))
        props[name] = fsa.CompositeShellProperty(lam)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "properties": props}})
    calc = InertialForcesCalculator([0.0, 0.0, 1.0])
    coords_d = torch.from_numpy(mesh.coords).to(asm.device)

    def step():
        K = asm.assemble_stiffness_matrix(recompute=True)
        M = asm.assemble_mass_matrix(recompute=True)
        lumped = M.lump_rows()
        F = calc.compute_centrifugal_force(coords_d, lumped[0::6].contiguous(), 0.79)
        return K, M, F

    for _ in range(3):
        K, M, F = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        K, M, F = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    E = mesh.element_count
    st = asm.fused_plan().stats()
    return {"workload": f"tapered tube 128 x 512 = {E} MITC4+ laminate elements, {len(asm._prop_table)} property rows, "
                        f"{int((asm._prop_table['flags'] & 1).sum())} with membrane-bending coupling",
            "ms_per_step": ms, "elements_per_s": E / (ms * 1e-3), "nnz": int(K.nnz),
            "fused_plan": {k: st[k] for k in ("patches", "stage1_recompute", "stage2_lane_fill")},
            "what": "K and M (fused kernel, two launches), row-sum lumping, centrifugal nodal loads"}


def mass_leg(torch, steps=10):
    """Consistent mass matrix of the 1.25 M-element strip: the fused kernel with FS_FUSED_MASS
    (one launch, no element mass matrices in memory) beside element kernel + assembly kernel; bitwise the same values."""
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    lams = c5_laminates()
    mesh = meshes.jittered_plate(STRIP_NX, STRIP_NY, cell=1e-3, seed=11, n_sets=16)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL,
                                            "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}})
    asm.fused_plan()
    E = mesh.element_count
    f_ms = avg_ms(torch, asm.assemble_mass_fused, steps)

    def two():
        asm.refresh_element_matrices(want_K=False, want_M=True, check=False)
        return asm.assemble_mass_matrix(recompute=False, fused=False)

    t_ms = avg_ms(torch, two, steps)
    Mf, M2 = asm.assemble_mass_fused(), two()
    same = bool((Mf.data.view(torch.int64) == M2.data.view(torch.int64)).all())
    gbs = E * BYTES_PER_ELEM / (f_ms * 1e-3) / 1e9
    return {"workload": f"{STRIP_NX}x{STRIP_NY}={E} element strip, consistent mass matrix",
            "fused_ms": f_ms, "two_kernel_ms": t_ms, "elements_per_s": E / (f_ms * 1e-3), "bitwise_equal_to_two_kernel_route": same,
            "roofline": {"kernel": "fused_quad_k_kernel<FS_OP_M>", "kernel_ms": f_ms, "bound": "hbm", "achieved": gbs,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "bytes_per_element": BYTES_PER_ELEM}}


def newton_leg(torch, steps=10):
    """configs[3]: geometric-nonlinear MITC4+ cantilever, tangent K_T and f_int re-assembled every Newton
    iteration (400 x 40 strip, E = 1.2e6, nu = 0, t = 0.1, u_k = k * 1e-3 * smooth field), and the
    same re-assembly on a 1.25 M-element plate.  Timed on its own, outside `value`."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
    out = {}
    for tag, nx, ny in (("c4_400x40", 400, 40), ("plate_400x3125", STRIP_NX, STRIP_NY)):
        mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * ny / 40, nx, ny)
        asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat,
                                                "thickness": 0.1, "nonlinear": True}})
        xs = torch.from_numpy(mesh.coords[:, 0]).to(asm.device)
        field = torch.zeros((mesh.node_count, 6), dtype=torch.float64, device=asm.device)
        field[:, 2] = xs ** 2 / 100.0
        field[:, 4] = -2.0 * xs / 100.0
        field = field.reshape(-1)
        asm.assemble_tangent_and_forces(field * 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(1, steps + 1):
            asm.assemble_tangent_and_forces(field * (k * 1e-3))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        step = asm.capture_tangent_and_forces()          # the same re-assembly replayed as one HIP graph
        step(field * 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(1, steps + 1):
            step(field * (k * 1e-3))
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t0) / steps * 1e3
        E = mesh.element_count
        # the route this one replaced: element kernel on the cached linear blocks + descriptor assembly + vector scatter
        asm.newton_on_assembled = False
        asm.assemble_tangent_and_forces(field * 1e-3)
        old_ms = avg_ms(torch, lambda: asm.assemble_tangent_and_forces(field * 2e-3), steps)
        asm.newton_on_assembled = True
        out[tag] = {"elements": E, "ms_per_reassembly": ms, "reassemblies_per_s": 1e3 / ms,
                    "elements_per_s": E / (ms * 1e-3), "ms_per_reassembly_hip_graph": gms,
                    "element_block_route_ms_per_reassembly": old_ms}
        if E > 1_000_000:      # per-kernel rooflines of the Newton path and of the kernels of the element-block route
            u = (field * 1e-3).contiguous()
            k_lin = asm._linear_csr()
            nnz = k_lin.numel()
            kt1_ms = avg_ms(torch, lambda: asm._tangent_on_assembled(u, k_lin), 10)
            asm.tangent_flags = 1
            kt_ms = avg_ms(torch, lambda: asm._tangent_on_assembled(u, k_lin), 10)
            asm.tangent_flags = 0
            f_ms = avg_ms(torch, lambda: asm._forces_on_assembled(u, k_lin), 10)
            # K_T: coordinates, node ids, u, id and 64-byte plan records per lane in; the packed blocks I <= J of K in;
            # all values out
            kt_b = E * (96 + 16 + 48 + 4) + 64 * asm._fplan_d["recs"].numel() // 64 + 8 * 36 * asm._fplan_d["n_upper_blocks"] + 8 * nnz
            f_b = 8 * nnz + E * (96 + 32 + 48 + 4 + 2 * 192) + asm.dofs_count * 8 * 3
            nl_ms = avg_ms(torch, lambda: asm._nl_pass(u, True, True), 5)
            nb = 96 + 4 + 192 + 2880 + 2880 + 192        # coordinates, id, u_e, linear blocks in; K_T blocks, f_int out
            m_ms = avg_ms(torch, lambda: asm.refresh_element_matrices(want_K=False, want_M=True, check=False), 5)
            mb = 96 + 4 + 4 + 2880
            roof = lambda ms_, by: {"kernel_ms": ms_, "bound": "hbm", "achieved": by / (ms_ * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                    "unit": "GB/s", "frac": by / (ms_ * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes": by}
            out["kernels"] = {
                "fused_quad_dkt_kernel<32> (K_T = K_csr + corrections, two workgroups per CU)": roof(kt_ms, kt_b),
                "fused_quad_dkt_kernel<64> (one workgroup per CU)": roof(kt1_ms, kt_b),
                "f_int: mitc4_nl_tiled_kernel (nonlinear part) + fs_vec_gather_sum + blk_matvec_kernel<6>": roof(f_ms, f_b),
                "mitc4_nl_tiled_kernel (+ fs_gather_dofs), element-block route": {
                    "kernel_ms": nl_ms, "bound": "hbm", "achieved": E * nb / (nl_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": E * nb / (nl_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_element": nb},
                "mitc4_m_tiled_kernel": {
                    "kernel_ms": m_ms, "bound": "hbm", "achieved": E * mb / (m_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": E * mb / (m_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_element": mb}}
        del asm
    out["what"] = "K_T(u_k) and f_int(u_k) re-assembled per Newton iterate on the ASSEMBLED linear matrix (kept once): " \
                  "K_T = K_csr + element-order sums of the geometric corrections in one fused launch, f_int = K_csr u + " \
                  "nonlinear membrane forces; element_block_route_*: gather u_e, TL kernel on the cached element blocks, " \
                  "CSR assembly, vector scatter (the route of rounds 1-2)"
    return out


def git_commit_of(path):
    try:
        return subprocess.run(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, capture_output=True,
                              text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def plate_run(torch, dist, rank, world, nx, ny, steps, warmup, distributed, return_matrix=False):
    """Build the assembler for an nx x ny piece of the plate (x offset rank * nx), run warmup + `steps` timed steps of
    assemble_stiffness_matrix(recompute=True) bracketed by barrier + synchronize, return the measurements
    (return_matrix=True: also the last step's matrix and the assembler, for tests/test_gpu_dist.py)."""
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    lams = c5_laminates()
    torch.cuda.reset_peak_memory_stats()
    log(f"plate {nx}x{ny}: mesh")
    t0 = time.perf_counter()
    # x-bands of 25 columns of the GLOBAL plate cycling through the 16 laminates: every strip is its part of the plate
    mesh = meshes.jittered_plate(nx, ny, cell=1e-3, seed=11, n_sets=16, x_offset_cells=rank * nx, band=BAND)
    names = mesh.set_names
    model = {"elements": {"element_family": ElementFamily.SHELL, "properties": {names[i]: lams[i] for i in range(16)}}}
    t1 = time.perf_counter()
    log(f"plate {nx}x{ny}: assembler (symbolic phase, first K)")
    if distributed:
        from fem_shell_amd.dist import DistributedAssembler
        asm = DistributedAssembler(mesh, model, strip=(rank, world, nx, ny), cache_blocks=False,
                                   overlap=os.environ.get("FEMSHELL_DIST_OVERLAP", "1") != "0")
        local = asm.local
    else:
        # cache_blocks=False: no element stiffness blocks in memory (the fused kernel never reads them)
        asm = local = MeshAssembler(mesh, model, cache_blocks=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if not local.can_fuse:
        raise SystemExit("bench: the plate must run through the fused kernel")
    if local._K_blk is not None:
        raise SystemExit("bench: cache_blocks=False must not leave element blocks in memory")
    log(f"plate {nx}x{ny}: fused patch plan")
    if not local.fused_ready():
        raise SystemExit(f"bench: no fused plan for the plate: {local._fplan_error}")
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    E = mesh.element_count
    log(f"plate {nx}x{ny}: timing")

    def barrier():
        if distributed:
            asm.finish()               # the last step's interface exchange (side stream) joins here
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    K = None
    # Host-side flow control: the host enqueues a step in ~0.1 ms and would run the whole loop ahead of the GPU.  With the
    # interface exchange on a side stream every step's value array stays reserved until ITS exchange has run, so a host that is
    # 20 steps ahead holds 20 arrays (13 GB each at N = 2).  The host therefore stays at most two steps ahead (it waits for
    # the event of the step before the previous one; the GPU queue never runs dry).
    import collections
    inflight = collections.deque()

    def throttle():
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > 2:
            inflight.popleft().synchronize()

    # Steady clocks before the W warm-up steps: after an idle period the first ~35 ms of kernel time run up to 20 % slower
    # (launch by launch in profiles/r03_launch_ramp.txt: 1.38 -> 1.55 -> 1.28 ms on the strip).  Three warm-up steps of the
    # 10 M plate cover that, three of a strip (1.3 ms each) do not, and the curve over N would compare a warm GPU at N = 1
    # with cold ones at N = 8.  So the same step runs untimed for about 100 ms first; the count depends on the strip size
    # only (the same on every rank: the interface exchanges pair up) and is reported as config.prewarm_steps.
    prewarm = 0 if os.environ.get("FEMSHELL_BENCH_NO_PREWARM") else min(100, int(np.ceil(0.1 / (E / 9.0e8))))
    for i in range(prewarm):
        K = None
        K = asm.assemble_stiffness_matrix(recompute=True)
        if i % 8 == 7:                 # (untimed: keep the side-stream exchanges and the value arrays they hold from piling up)
            if distributed:
                asm.finish()
            torch.cuda.synchronize()
    for _ in range(warmup):
        K = None                       # (the previous step's values go back to the allocator before the next 26 GB are taken)
        K = asm.assemble_stiffness_matrix(recompute=True)
        throttle()
    barrier()
    inflight.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0s = time.perf_counter()
    e0.record()
    for _ in range(steps):
        K = None
        K = asm.assemble_stiffness_matrix(recompute=True)
        throttle()
    e1.record()
    barrier()
    dt = time.perf_counter() - t0s
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stream_ms = e0.elapsed_time(e1) / steps            # launch stream only: the fused kernel, back to back
    nnz = int(K.nnz)
    if not distributed:
        K = None                                       # (not kept beside the next measurement's value arrays)
    # the kernel alone, HIP events on its stream, nothing else in flight (no exchange): the launch of a step, status words
    # written (they are among the algorithmic bytes; the 40 MB fill that zeroes them first is inside the events)
    st = torch.zeros(E, dtype=torch.int32, device="cuda")
    k_ms = avg_ms(torch, lambda: local.assemble_stiffness_fused(status=st), max(3, min(steps, 10)))
    del st
    res = {"E": E, "dt": dt, "kernel_ms": k_ms, "nx": nx, "ny": ny, "launch_stream_ms_per_step": stream_ms, "nnz": nnz,
           "prewarm_steps": prewarm,
           "coupled_frac": float(np.mean((local._prop_table["flags"][local._prop_id] & 1) != 0)),
           "setup_s": {"mesh": t1 - t0, "assembler_symbolic_and_first_K": t2 - t1, "fused_plan": t3 - t2},
           "plan": local.fused_plan().stats(),
           "device_memory_GB": torch.cuda.max_memory_allocated() / 1e9}
    if distributed:
        # the exchange on its own (pack -> grouped send/recv -> add), every rank at the same time
        barrier()
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record()
        for _ in range(5):
            asm._exchange_on_current_stream(K)
        x1.record()
        torch.cuda.synchronize()
        res["exchange_ms"] = x0.elapsed_time(x1) / 5
        res["interface_exchange_MB"] = sum(b.numel() for b in list(asm._sbuf.values()) + list(asm._rbuf.values())) * 8 / 1e6
        res["overlap"] = asm.overlap
        res["setup_s"]["distributed_plan"] = asm.setup_s
    local.check_status()      # the status words the timed steps wrote
    if return_matrix:
        K = asm.assemble_stiffness_matrix(recompute=True)   # (the exchange timing above added the interface rows to K again)
        if distributed:
            asm.finish()
        return res, K, asm
    del K, asm, local
    torch.cuda.empty_cache()
    return res


def full_workload_matches(args, r):
    """The PMC children ran the plate unless --strip-only; if this process then fell back to the strip for lack of
    memory, their figure is for another workload and is not reported as `traffic`."""
    return r["E"] == (STRIP_NX * STRIP_NY if args.strip_only else FULL_NX * FULL_NY)


def roofline_objects(r, ms_per_step):
    """`roofline` for the dominant (only) kernel of the step and `roofline_step` for the whole step, both priced with the
    ALGORITHMIC bytes (SURVEY.md 8d fused path) and flops (bench/FLOPS.md) per element."""
    E = r["E"]
    flop = FLOP_PER_ELEM + r["coupled_frac"] * (FLOP_PER_ELEM_COUPLED - FLOP_PER_ELEM)
    prof = None
    try:
        pj = os.path.join(ROOT, "profiles", "pmc_summary.json")
        pm = json.load(open(pj)).get("fused_quad_k_kernel")
        if pm:
            prof = {"hbm_bytes_per_launch": pm.get("hbm_bytes_per_launch"), "elements": pm.get("elements"),
                    "file": "profiles/pmc_summary.json", "commit": git_commit_of("profiles/pmc_summary.json"),
                    "note": "rocprofv3 --pmc passes of an earlier run of this kernel on the 1.25 M strip, NOT observed in this run"}
    except Exception:
        prof = None

    def obj(ms):
        gbs = E * BYTES_PER_ELEM / (ms * 1e-3) / 1e9
        tfl = E * flop / (ms * 1e-3) / 1e12
        return gbs, tfl

    gbs, tfl = obj(r["kernel_ms"])
    roof = {"kernel": "fused_quad_k_kernel", "kernel_ms": r["kernel_ms"], "bound": "hbm", "achieved": gbs,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
            "traffic_from_profiles": prof, "bytes_per_element": BYTES_PER_ELEM, "elements_per_launch": E,
            "valu_fp64": {"achieved": tfl, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": tfl / PEAK_FP64_TFLOPS,
                          "flop_per_element": flop, "source": "bench/FLOPS.md"},
            "note": "at 2 708 B and ~20 kflop per element the HBM roofline (0.34 us per 1 000 elements) lies above the fp64 "
                    "VALU roofline (0.26 us), so HBM is the bound quoted; the kernel itself is limited by VALU issue and "
                    "LDS latency (DESIGN.md 4.5)"}
    sg, stf = obj(ms_per_step)
    step = {"ms_per_step": ms_per_step, "bytes_per_element": BYTES_PER_ELEM, "flop_per_element": flop,
            "hbm": {"achieved": sg, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": sg / PEAK_HBM_GBS},
            "valu_fp64": {"achieved": stf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": stf / PEAK_FP64_TFLOPS}}
    return roof, step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the configs[0..3] / HEXA8 legs")
    ap.add_argument("--strip-only", action="store_true", help="N = 1: time the 1.25 M strip instead of the 10 M plate")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    traffic = (None, None)
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    if world == 1 and not args.pmc_child and not args.no_pmc and not under_profiler:
        # before this process touches the GPU: the children need the memory the parent is about to take
        log("PMC traffic passes (rocprofv3 children)")
        traffic = pmc_traffic(full=not args.strip_only)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    backend = os.environ.get("FEMSHELL_DIST_BACKEND", "nccl")   # "gloo": one-GPU rehearsal only
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    strip = None
    if world > 1:
        if FULL_NX % world:
            raise SystemExit(f"--gpus {world}: the plate's {FULL_NX} element columns do not divide into {world} strips")
        snx = FULL_NX // world
        r = plate_run(torch, dist, rank, world, snx, FULL_NY, args.steps, args.warmup, True)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "elements": r["E"], "kernel_ms": r["kernel_ms"],
                                          "exchange_ms": r["exchange_ms"],
                                          "launch_stream_ms_per_step": r["launch_stream_ms_per_step"],
                                          "coupled_frac": r["coupled_frac"],
                                          "setup_s": r["setup_s"], "interface_exchange_MB": r["interface_exchange_MB"]})
        workload = (f"C5 synthetic jittered MITC4+ laminate plate, {FULL_NX}x{FULL_NY}={r['E'] * world} elements cut into "
                    f"{world} x-strips of {snx}x{FULL_NY}={r['E']} elements (one per GPU), 16 laminates in x-bands of {BAND} "
                    f"columns, {r['coupled_frac']:.0%} of this rank's elements coupled; step = fused evaluate-and-assemble "
                    "kernel (status words written) + RCCL interface-row exchange")
    else:
        free = torch.cuda.mem_get_info()[0]
        full = not args.strip_only and free >= 120e9
        if args.pmc_child:      # under rocprofv3 --pmc: the workload of `value` only, nothing printed
            nx, ny = (FULL_NX, FULL_NY) if full else (STRIP_NX, STRIP_NY)
            plate_run(torch, dist, 0, 1, nx, ny, args.steps, args.warmup, False)
            return
        if not args.strip_only:
            strip = plate_run(torch, dist, 0, 1, STRIP_NX, STRIP_NY, args.steps, args.warmup, False)
        if full:
            r = plate_run(torch, dist, 0, 1, FULL_NX, FULL_NY, args.steps, args.warmup, False)
        else:
            r = strip or plate_run(torch, dist, 0, 1, STRIP_NX, STRIP_NY, args.steps, args.warmup, False)
        nx, ny = (FULL_NX, FULL_NY) if full else (STRIP_NX, STRIP_NY)
        workload = (f"C5 synthetic jittered MITC4+ laminate plate, {nx}x{ny}={r['E']} elements on one GPU"
                    + (" (configs[4]'s full shell plate)" if full else f" (one x-strip: {free / 1e9:.0f} GB free or --strip-only)")
                    + f", 16 laminates striped by x-band, {r['coupled_frac']:.0%} of the elements coupled; "
                    "step = one launch of the fused evaluate-and-assemble kernel (status words written)")
        per_rank = None

    if rank == 0:
        E = r["E"]
        total_E = E * world
        ms_per_step = r["dt"] / args.steps * 1e3
        value = total_E * args.steps / r["dt"]
        roof, roof_step = roofline_objects(r, ms_per_step)
        if traffic[0] is not None and full_workload_matches(args, r):
            roof["traffic"] = traffic[0]
            roof["traffic_over_algorithmic"] = traffic[0] / (r["E"] * BYTES_PER_ELEM)
        if traffic[1] is not None:
            roof["traffic_measurement"] = traffic[1]
        line = {
            "metric": "MITC4+ element stiffness matrices assembled into CSR per second (fp64)",
            "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "elements_per_gpu": E, "nnz_per_gpu": r["nnz"],
                       "partition": f"x-strips x{world}" if world > 1 else "none",
                       "prewarm_steps": r["prewarm_steps"],      # untimed, before the W warm-up steps (steady clocks: plate_run)
                       "setup_s_rank0": r["setup_s"], "device_memory_GB_rank0": r["device_memory_GB"],
                       "fused_plan": {k: r["plan"][k] for k in ("patches", "elements_per_patch", "stage1_recompute",
                                                                 "stage2_lane_fill", "record_bytes_per_element")},
                       "exchange": ("RCCL grouped send/recv on a side stream, overlapped with the next step's kernel"
                                    if world > 1 and r["overlap"] else ("RCCL grouped send/recv" if world > 1 else "none"))},
            "roofline": roof,
            "roofline_step": roof_step,
        }
        if per_rank is not None:
            line["per_rank"] = per_rank
            km = [q["kernel_ms"] for q in per_rank]
            line["load_balance_kernel_ms_max_over_min"] = max(km) / min(km)
            # step time not covered by this rank's kernel: exchange (and waiting for neighbours) that the overlap did
            # not hide; 0 when the side-stream exchange disappears under the next step's kernel
            line["exchange_exposed_ms"] = max(0.0, ms_per_step - max(km))
            line["multi_gpu_note"] = ("RCCL transport unmeasured on hardware by the builder (one-GPU boxes only); this line "
                                      "is its first multi-GPU run")
        if strip is not None and strip is not r:
            sms = strip["dt"] / args.steps * 1e3
            sroof, sstep = roofline_objects(strip, sms)
            line["one_rank_share_at_n8"] = {"workload": f"{STRIP_NX}x{STRIP_NY}={strip['E']} element x-strip (rank 0's share of the same plate at N = 8)",
                                          "value": strip["E"] * args.steps / strip["dt"], "unit": "elements/s",
                                          "ms_per_step": sms, "kernel_ms": strip["kernel_ms"], "nnz": strip["nnz"],
                                          "roofline_frac_hbm": sroof["frac"], "setup_s": strip["setup_s"]}
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline")
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        if world == 1 and not args.no_legs:
            log("measured peaks")
            mp = measured_peaks(torch)
            line["measured_peaks"] = mp
            roof["frac_of_measured_hbm_copy"] = roof["achieved"] / mp["hbm_copy_GBs"]
            roof["valu_fp64"]["frac_of_measured_fma_rate"] = roof["valu_fp64"]["achieved"] / mp["fp64_fma_4_waves_per_simd_TFLOPs"]
            for key, leg in (("mass", mass_leg), ("hex8", hex8_leg), ("newton", newton_leg), ("c1_scordelis_lo", c1_leg),
                             ("c2_hemisphere", c2_leg), ("c3_blade_like", c3_leg)):
                log(f"leg {key}")
                line[key] = leg(torch)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
