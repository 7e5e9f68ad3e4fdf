"""GPU: DistributedAssembler with 2 ranks sharing the one GPU of the box (gloo transport; the
8-GPU RCCL run is the driver's).  Owned rows must equal the serial single-GPU assembly."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, nx, ny, q, overlap=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fem_shell_amd import meshes
        from fem_shell_amd.assembler import MeshAssembler
        from fem_shell_amd.dist import DistributedAssembler
        from fem_shell_amd.elements import ElementFamily
        from tests import util
        props = util.standard_props()
        model = {"elements": {"element_family": ElementFamily.SHELL,
                              "properties": {f"band_{i}": props[i] for i in range(8)}}}
        g = meshes.jittered_plate(nx * world, ny, n_sets=8)
        ref = MeshAssembler(g, model).assemble_stiffness_matrix().to_scipy()
        loc = meshes.jittered_plate(nx, ny, n_sets=8, x_offset_cells=rank * nx)
        # the strip must use the GLOBAL band of each element
        loc.set_id = g.set_id.reshape(ny, nx * world)[:, rank * nx:(rank + 1) * nx].reshape(-1).copy()
        da = DistributedAssembler(loc, model, strip=(rank, world, nx, ny), overlap=overlap)
        A = da.assemble_stiffness_matrix(recompute=True)
        if overlap:       # a second step while the first exchange may still be in flight, then join
            da.refresh_element_matrices(want_K=True, want_M=False, check=False)
            A = da.assemble_stiffness_matrix(recompute=False)
            da.finish()
        ip, ix, data = A.indptr.cpu().numpy(), A.indices.cpu().numpy(), A.data.cpu().numpy()
        X = da.dplan.nodes_g
        worst = 0.0
        for r in np.flatnonzero(da.owned_rows):
            gr = 6 * X[r // 6] + r % 6
            cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
            assert np.array_equal(cols, ref.indices[ref.indptr[gr]:ref.indptr[gr + 1]])
            worst = max(worst, np.abs(data[ip[r]:ip[r + 1]] - ref.data[ref.indptr[gr]:ref.indptr[gr + 1]]).max())
        assert worst <= 1e-13 * np.abs(ref.data).max(), worst
        q.put((rank, int(da.owned_rows.sum()), None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (2, True), (3, True)])
def test_ranks_share_one_gpu(world, overlap):
    """world = 3: the middle strip has a neighbour on either side (it ships rows to the lower rank and receives the
    upper rank's)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 9, 7, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 6 * (9 * world + 1) * 8


def _worker_newton(rank, world, port, nx, ny, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fem_shell_amd as fsa
        from fem_shell_amd import meshes
        from fem_shell_amd.assembler import MeshAssembler
        from fem_shell_amd.dist import DistributedAssembler
        from fem_shell_amd.elements import ElementFamily
        mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.3, 1.0)
        model = {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1, "nonlinear": True}}
        g = meshes.jittered_plate(nx * world, ny, cell=0.1, n_sets=1)
        ug = 1e-3 * np.random.default_rng(5).standard_normal(g.node_count * 6)       # the same field on every rank
        serial = MeshAssembler(g, model)
        Kg, fg = serial.assemble_tangent_and_forces(ug)
        Kg, fg = Kg.to_scipy(), fg.cpu().numpy()
        loc = meshes.jittered_plate(nx, ny, cell=0.1, n_sets=1, x_offset_cells=rank * nx)
        da = DistributedAssembler(loc, model, strip=(rank, world, nx, ny), overlap=True)
        X = da.dplan.nodes_g
        ul = ug.reshape(-1, 6)[X].reshape(-1)
        A, f = da.assemble_tangent_and_forces(ul)
        f_only = da.assemble_internal_forces(ul)
        A_only = da.assemble_tangent_stiffness(ul)
        da.finish()
        ip, ix, data = A.indptr.cpu().numpy(), A.indices.cpu().numpy(), A.data.cpu().numpy()
        # (f_int alone takes its nonlinear part from the element kernel + vector scatter, the combined call from the
        # tangent kernel's own sums: the same terms, rounding-level apart)
        assert bool((A_only.data == A.data).all()) and float((f_only - f).abs().max()) <= 1e-12 * float(f.abs().max())
        f = f.cpu().numpy()
        worstK = worstF = 0.0
        for r in np.flatnonzero(da.owned_rows):
            gr = 6 * X[r // 6] + r % 6
            cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
            assert np.array_equal(cols, Kg.indices[Kg.indptr[gr]:Kg.indptr[gr + 1]])
            worstK = max(worstK, np.abs(data[ip[r]:ip[r + 1]] - Kg.data[Kg.indptr[gr]:Kg.indptr[gr + 1]]).max())
            worstF = max(worstF, abs(f[r] - fg[gr]))
        assert worstK <= 1e-13 * np.abs(Kg.data).max() and worstF <= 1e-12 * np.abs(fg).max(), (worstK, worstF)
        q.put((rank, int(da.owned_rows.sum()), None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_distributed_newton_reassembly():
    """K_T(u) and f_int(u) over 3 ranks sharing the GPU: matrix interface exchange (side stream) + vector exchange;
    owned rows equal the serial re-assembly."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_worker_newton, args=(r, world, port, 6, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 6 * (6 * world + 1) * 6


def _worker_solid(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from types import SimpleNamespace
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fem_shell_amd import meshes
        from fem_shell_amd.assembler import MeshAssembler
        from fem_shell_amd.dist import DistributedAssembler
        from fem_shell_amd.elements import ElementFamily
        from tests import util
        nx, ny, nz = 3 * world, 3, 2
        ids = np.arange((nx + 1) * (ny + 1) * (nz + 1)).reshape(nz + 1, ny + 1, nx + 1)
        Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
        coords = np.stack([X, Y, Z], -1).reshape(-1, 3) + 0.05 * np.random.default_rng(4).uniform(-1, 1, (ids.size, 3))
        n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
        hexa = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)], -1)
        model = {"elements": {"element_family": ElementFamily.SOLID, "material": util.t300()}}
        body = SimpleNamespace(value=np.array([0.2, -1.0, -9.81]))
        serial = MeshAssembler(meshes.ArrayMesh(coords, hexa.reshape(-1, 8)), model)
        Kg = serial.assemble_stiffness_matrix().to_scipy()
        fg = serial.assemble_load_vector(body).cpu().numpy()
        mine = hexa[:, :, 3 * rank:3 * rank + 3].reshape(-1, 8)          # this rank's x-slab of bricks
        l2g = np.unique(mine)
        loc = meshes.ArrayMesh(coords[l2g], np.searchsorted(l2g, mine))
        da = DistributedAssembler(loc, model, l2g=l2g, overlap=True)
        A = da.assemble_stiffness_matrix(recompute=True)
        f = da.assemble_load_vector(body)
        da.finish()
        ip, ix, data, f = A.indptr.cpu().numpy(), A.indices.cpu().numpy(), A.data.cpu().numpy(), f.cpu().numpy()
        Xg = da.dplan.nodes_g
        worstK = worstF = 0.0
        for r in np.flatnonzero(da.owned_rows):
            gr = 3 * Xg[r // 3] + r % 3
            cols = 3 * Xg[ix[ip[r]:ip[r + 1]] // 3] + ix[ip[r]:ip[r + 1]] % 3
            assert np.array_equal(cols, Kg.indices[Kg.indptr[gr]:Kg.indptr[gr + 1]])
            worstK = max(worstK, np.abs(data[ip[r]:ip[r + 1]] - Kg.data[Kg.indptr[gr]:Kg.indptr[gr + 1]]).max())
            worstF = max(worstF, abs(f[r] - fg[gr]))
        assert worstK <= 1e-13 * np.abs(Kg.data).max() and worstF <= 1e-13 * np.abs(fg).max(), (worstK, worstF)
        q.put((rank, int(da.owned_rows.sum()), None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_distributed_solid_block():
    """configs[4]'s solid block is element-partitioned too: HEXA8 x-slabs over 2 ranks (3 dofs per node), K and the
    body-load vector; owned rows equal the serial assembly."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker_solid, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 3 * (3 * world + 1) * 4 * 3


def _worker_blocks(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fem_shell_amd as fsa
        from fem_shell_amd import meshes
        from fem_shell_amd.assembler import MeshAssembler
        from fem_shell_amd.dist import DistributedAssembler, local_block
        from fem_shell_amd.elements import ElementFamily
        g = meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 10, triangular=True)    # 200 MITC3+ triangles
        model = {"elements": {"element_family": ElementFamily.SHELL, "thickness": 0.04,
                              "material": fsa.IsotropicMaterial("s", 6.825e7, 0.3, 1.0)}}
        serial = MeshAssembler(g, model)
        Kg, Mg = serial.assemble_stiffness_matrix().to_scipy(), serial.assemble_mass_matrix().to_scipy()
        loc, l2g = local_block(g, rank, world)
        assert loc.element_count in (66, 67) and np.array_equal(loc.coords, g.coords[l2g])
        da = DistributedAssembler(loc, model, l2g=l2g)
        X = da.dplan.nodes_g
        for A, ref in ((da.assemble_stiffness_matrix(recompute=True), Kg), (da.assemble_mass_matrix(recompute=True), Mg)):
            ip, ix, data = A.indptr.cpu().numpy(), A.indices.cpu().numpy(), A.data.cpu().numpy()
            worst = 0.0
            for r in np.flatnonzero(da.owned_rows):
                gr = 6 * X[r // 6] + r % 6
                cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
                assert np.array_equal(cols, ref.indices[ref.indptr[gr]:ref.indptr[gr + 1]])
                worst = max(worst, np.abs(data[ip[r]:ip[r + 1]] - ref.data[ref.indptr[gr]:ref.indptr[gr + 1]]).max())
            assert worst <= 1e-13 * np.abs(ref.data).max(), worst
        q.put((rank, int(da.owned_rows.sum()), None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_distributed_element_blocks_of_any_mesh():
    """local_block(): contiguous element blocks of a triangulated hemisphere patch over 3 ranks (interfaces are not
    straight lines: a block boundary cuts through a row of triangles); K and M owned rows equal the serial result."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_worker_blocks, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 6 * 121


def _worker_nccl(port, q):
    """One rank, backend nccl (= RCCL): the process-group calls of the bench's N > 1 path on device tensors."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import bench
        r = bench.plate_run(torch, dist, 0, 1, 48, 40, 3, 1, True)
        tt = torch.tensor([r["dt"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        per_rank = [None]
        dist.all_gather_object(per_rank, {"rank": 0, "exchange_ms": r["exchange_ms"]})
        q.put((r["E"], r["nnz"], r["overlap"], per_rank[0]["rank"], None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((0, 0, None, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_nccl_backend_world_1():
    """The RCCL transport needs one GPU per rank, so a one-GPU box can only run it with ONE rank: this covers the
    process-group set-up of `bench.py --gpus N` (init with device_id, the bounding-box all_gather on device tensors,
    all_reduce, barrier, all_gather_object, side-stream bookkeeping with no neighbour); the send / recv pairs themselves
    stay unmeasured on hardware."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl, args=(port, q))
    p.start()
    E, nnz, overlap, rank, err = q.get(timeout=240)
    p.join(30)
    assert err is None, err
    assert E == 48 * 40 and nnz == 36 * (3 * 49 - 2) * (3 * 41 - 2) and overlap is True and rank == 0


def _worker_bench(rank, world, port, nx, ny, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from fem_shell_amd import meshes
        from fem_shell_amd.assembler import MeshAssembler
        from fem_shell_amd.elements import ElementFamily
        res, A, da = bench.plate_run(torch, dist, rank, world, nx, ny, 2, 1, True, return_matrix=True)
        da.finish()
        torch.cuda.synchronize()
        assert res["E"] == nx * ny and res["exchange_ms"] > 0 and da.local._K_blk is None
        # the serial plate of the same bench workload (global laminate bands), through the same fused kernel
        lams = bench.c5_laminates()
        g = meshes.jittered_plate(nx * world, ny, cell=1e-3, seed=11, n_sets=16, band=bench.BAND)
        ser = MeshAssembler(g, {"elements": {"element_family": ElementFamily.SHELL,
                                             "properties": {g.set_names[i]: lams[i] for i in range(16)}}}, cache_blocks=False)
        ref = ser.assemble_stiffness_matrix(recompute=True).to_scipy()
        assert ser._K_blk is None
        ip, ix, data = A.indptr.cpu().numpy(), A.indices.cpu().numpy(), A.data.cpu().numpy()
        X = da.dplan.nodes_g
        worst = 0.0
        for r in np.flatnonzero(da.owned_rows):
            gr = 6 * X[r // 6] + r % 6
            cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
            assert np.array_equal(cols, ref.indices[ref.indptr[gr]:ref.indptr[gr + 1]])
            worst = max(worst, np.abs(data[ip[r]:ip[r + 1]] - ref.data[ref.indptr[gr]:ref.indptr[gr + 1]]).max())
        assert worst <= 1e-13 * np.abs(ref.data).max(), worst
        q.put((rank, int(da.owned_rows.sum()), None))
    except BaseException:  # noqa: BLE001  (SystemExit of the bench included)
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bench_strong_scaling_strips_two_ranks():
    """bench.py --gpus N cuts ONE plate into N x-strips (strong scaling).  The bench's own plate_run(distributed=True)
    with 2 ranks on this GPU (gloo): owned rows of both strips equal the serial plate of the same workload (global
    laminate bands: a band boundary falls inside each strip), no element blocks in memory (cache_blocks=False)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, nx, ny = 2, 40, 21
    procs = [ctx.Process(target=_worker_bench, args=(r, world, port, nx, ny, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 6 * (nx * world + 1) * (ny + 1)
