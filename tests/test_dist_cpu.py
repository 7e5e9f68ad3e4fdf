"""CPU rehearsal (gloo, world_size 2, 3, 4 and 8) of the multi-GPU path: partition, ghost-column pattern
exchange and interface-row value exchange must reproduce the serial assembly on owned rows.
Element matrices come from the oracle and the kernels are emulated in NumPy (test infrastructure);
the exchanged index plans are exactly the ones DistributedAssembler uses on the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, ny, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fem_shell_amd as fsa
        from fem_shell_amd import meshes
        from fem_shell_amd.dist import DistributedPlan
        from fem_shell_amd.symbolic import BlockPlan
        from oracle import oracle as O
        from tests import util
        props = fsa.build_prop_table(util.standard_props())
        # serial reference: the whole plate
        g = meshes.jittered_plate(nx * world, ny, n_sets=8)
        gpid = g.set_id.astype(np.int32)
        Kg, _ = O.mitc4_K_batch(g.coords[g.conn], gpid, props)
        gplan = BlockPlan(g.conn, g.node_count, 6)
        gdata = util.host_block_assemble(gplan, util.pack_symblk(Kg))
        gip, gix = gplan.host_pattern()
        # this rank's strip, local numbering, global ids
        loc = meshes.jittered_plate(nx, ny, n_sets=8, x_offset_cells=rank * nx)
        gx = np.arange(nx + 1) + rank * nx
        l2g = (np.arange(ny + 1)[:, None] * (world * nx + 1) + gx[None, :]).reshape(-1)
        assert np.array_equal(loc.coords, g.coords[l2g])
        ecol = (np.arange(nx) + rank * nx)
        eg = (np.arange(ny)[:, None] * (world * nx) + ecol[None, :]).reshape(-1)     # global element ids
        dp = DistributedPlan(loc.conn, l2g, 6, coords=loc.coords)
        # bounding boxes: a strip only ever talks to the strips next to it, and only about its two edge columns
        assert dp.neighbors == [r for r in (rank - 1, rank + 1) if 0 <= r < world]
        assert dp.candidate_ids_sent <= 2 * len(dp.neighbors) * (ny + 1)
        Kl, _ = O.mitc4_K_batch(loc.coords[loc.conn], gpid[eg], props)
        assert dp.plan.tiled      # quad strips use the tiled element-matrix layout
        data = util.host_block_assemble(dp.plan, util.pack_symblk(Kl, tiled=dp.plan.tiled))
        data = dp.exchange_host(data)
        ip, ix = dp.plan.host_pattern()
        X = dp.nodes_g
        nchk = 0
        for I in np.flatnonzero(dp.owned):
            for k in range(6):
                r, gr = 6 * I + k, 6 * X[I] + k
                cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
                assert np.array_equal(cols, gix[gip[gr]:gip[gr + 1]]), (rank, I, k)     # pattern: exact
                ref = gdata[gip[gr]:gip[gr + 1]]
                got = data[ip[r]:ip[r + 1]]
                assert np.abs(got - ref).max() <= 1e-13 * np.abs(gdata).max(), (rank, I, k)
                nchk += 1
        # vector exchange (f_int / load vectors): element vectors summed per rank, interface rows exchanged
        rng = np.random.default_rng(100)
        fe_g = rng.standard_normal((g.conn.shape[0], 24))               # one "element force" per global element
        gdofs = (g.conn[:, :, None] * 6 + np.arange(6)).reshape(g.conn.shape[0], -1)
        f_ref = np.zeros(g.node_count * 6)
        np.add.at(f_ref, gdofs.reshape(-1), fe_g.reshape(-1))
        ldofs = (dp.conn_x[:, :, None] * 6 + np.arange(6)).reshape(loc.conn.shape[0], -1)
        f_loc = np.zeros(X.shape[0] * 6)
        np.add.at(f_loc, ldofs.reshape(-1), fe_g[eg].reshape(-1))
        f_loc = dp.exchange_vector_host(f_loc)
        for I in np.flatnonzero(dp.owned):
            assert np.abs(f_loc[6 * I:6 * I + 6] - f_ref[6 * X[I]:6 * X[I] + 6]).max() <= 1e-13 * np.abs(f_ref).max(), (rank, I)
        owned_g = torch.zeros(g.node_count, dtype=torch.int64)
        owned_g[torch.from_numpy(X[dp.owned])] = 1
        dist.all_reduce(owned_g)
        assert bool((owned_g == 1).all())          # every block row owned exactly once
        q.put((rank, nchk, None))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_distributed_plan_matches_serial(world):
    """world 8 = the C5 strip topology of bench.py --gpus 8: 8 x-strips, 7 interfaces."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 5 if world < 8 else 3, 4, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
        assert n > 0


def _worker_quadrants(rank, world, port, nx, ny, q, use_coords):
    """2 x 2 element quadrants: the centre node belongs to all four ranks, edge-interface nodes to two."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fem_shell_amd as fsa
        from fem_shell_amd import meshes
        from fem_shell_amd.dist import DistributedPlan
        from fem_shell_amd.symbolic import BlockPlan
        from oracle import oracle as O
        from tests import util
        props = fsa.build_prop_table(util.standard_props())
        g = meshes.jittered_plate(nx, ny, n_sets=8)
        gpid = g.set_id.astype(np.int32)
        Kg, _ = O.mitc4_K_batch(g.coords[g.conn], gpid, props)
        gplan = BlockPlan(g.conn, g.node_count, 6)
        gdata = util.host_block_assemble(gplan, util.pack_symblk(Kg))
        gip, gix = gplan.host_pattern()
        ex, ey = np.meshgrid(np.arange(nx), np.arange(ny))
        quadrant = ((ey >= ny // 2) * 2 + (ex >= nx // 2)).reshape(-1)
        eg = np.flatnonzero(quadrant == rank)                               # global element ids, ascending
        l2g = np.unique(g.conn[eg])
        conn = np.searchsorted(l2g, g.conn[eg])
        dp = DistributedPlan(conn, l2g, 6, coords=g.coords[l2g] if use_coords else None)
        assert len(dp.neighbors) == 3                                       # every quadrant touches the other three
        Kl, _ = O.mitc4_K_batch(g.coords[g.conn[eg]], gpid[eg], props)
        data = util.host_block_assemble(dp.plan, util.pack_symblk(Kl, tiled=dp.plan.tiled))
        data = dp.exchange_host(data)
        ip, ix = dp.plan.host_pattern()
        X = dp.nodes_g
        nchk = 0
        for I in np.flatnonzero(dp.owned):
            for k in range(6):
                r, gr = 6 * I + k, 6 * X[I] + k
                cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
                assert np.array_equal(cols, gix[gip[gr]:gip[gr + 1]]), (rank, I, k)
                assert np.abs(data[ip[r]:ip[r + 1]] - gdata[gip[gr]:gip[gr + 1]]).max() <= 1e-13 * np.abs(gdata).max()
                nchk += 1
        owned_g = torch.zeros(g.node_count, dtype=torch.int64)
        owned_g[torch.from_numpy(X[dp.owned])] = 1
        dist.all_reduce(owned_g)
        assert bool((owned_g == 1).all())
        q.put((rank, nchk, None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_coords", [True, False])
def test_distributed_plan_two_by_two_quadrants(use_coords):
    """use_coords=False: the fallback without coordinates (every rank's node list to every rank)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_quadrants, args=(r, 4, port, 6, 6, q, use_coords)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 6 * 49


def test_rcm_element_order_cuts_the_interface():
    """dist.local_block(order="rcm"): the reference's RCM renumbering (core/mesh/model.py:229-321) as the partition
    order.  On a plate whose elements are numbered at random, contiguous blocks of the input order share almost every
    node with every other block; blocks of the RCM order share a front."""
    from fem_shell_amd import meshes
    from fem_shell_amd.dist import local_block, rcm_element_order
    g = meshes.jittered_plate(40, 30, n_sets=1)
    rng = np.random.default_rng(0)
    shuffled = meshes.ArrayMesh(g.coords, g.conn[rng.permutation(g.conn.shape[0])])
    perm, new_of_old = rcm_element_order(shuffled.conn, shuffled.coords.shape[0])
    assert np.array_equal(np.sort(perm), np.arange(shuffled.conn.shape[0]))
    assert np.array_equal(np.sort(new_of_old), np.arange(shuffled.coords.shape[0]))
    key = new_of_old[shuffled.conn].min(axis=1)[perm]
    assert (np.diff(key) >= 0).all()                      # elements follow their smallest new node index

    def shared_nodes(order):
        parts = [local_block(shuffled, r, 4, order=order)[1] for r in range(4)]
        allc = np.concatenate(parts)
        return allc.shape[0] - np.unique(allc).shape[0], parts
    n_in, _ = shared_nodes(None)
    n_rcm, parts = shared_nodes("rcm")
    print("nodes held by more than one rank: input order", n_in, "RCM order", n_rcm)
    assert n_rcm * 8 < n_in and n_rcm < 4 * 3 * 45        # a few fronts of about one mesh line each
    # every element lands in exactly one block, with its connectivity intact
    tot = 0
    for r in range(4):
        loc, l2g = local_block(shuffled, r, 4, order="rcm")
        tot += loc.conn.shape[0]
        assert np.array_equal(np.sort(l2g), l2g)
    assert tot == shuffled.conn.shape[0]


def _worker_rcm(rank, world, port, q):
    """Randomly numbered plate cut into RCM blocks: plan + exchange reproduce the serial assembly on owned rows."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fem_shell_amd as fsa
        from fem_shell_amd import meshes
        from fem_shell_amd.dist import DistributedPlan, local_block, rcm_element_order
        from fem_shell_amd.symbolic import BlockPlan
        from oracle import oracle as O
        from tests import util
        props = fsa.build_prop_table(util.standard_props())
        g0 = meshes.jittered_plate(9, 7, n_sets=8)
        perm0 = np.random.default_rng(4).permutation(g0.conn.shape[0])
        g = meshes.ArrayMesh(g0.coords, g0.conn[perm0], g0.set_id[perm0], g0.set_names)
        gpid = g.set_id.astype(np.int32)
        Kg, _ = O.mitc4_K_batch(g.coords[g.conn], gpid, props)
        gplan = BlockPlan(g.conn, g.node_count, 6)
        gdata = util.host_block_assemble(gplan, util.pack_symblk(Kg))
        gip, gix = gplan.host_pattern()
        loc, l2g = local_block(g, rank, world, order="rcm")
        perm = rcm_element_order(g.conn, g.node_count)[0]
        E = g.conn.shape[0]
        eg = np.sort(perm[(E * rank) // world:(E * (rank + 1)) // world])
        assert np.array_equal(l2g[loc.conn], g.conn[eg])
        dp = DistributedPlan(loc.conn, l2g, 6, coords=loc.coords)
        Kl, _ = O.mitc4_K_batch(loc.coords[loc.conn], gpid[eg], props)
        data = util.host_block_assemble(dp.plan, util.pack_symblk(Kl, tiled=dp.plan.tiled))
        data = dp.exchange_host(data)
        ip, ix = dp.plan.host_pattern()
        X = dp.nodes_g
        nchk = 0
        for I in np.flatnonzero(dp.owned):
            for k in range(6):
                r, gr = 6 * I + k, 6 * X[I] + k
                cols = 6 * X[ix[ip[r]:ip[r + 1]] // 6] + ix[ip[r]:ip[r + 1]] % 6
                assert np.array_equal(cols, gix[gip[gr]:gip[gr + 1]]), (rank, I, k)
                # interface rows sum the ranks' partial sums: same terms, another association -- tolerance, not bits
                assert np.abs(data[ip[r]:ip[r + 1]] - gdata[gip[gr]:gip[gr + 1]]).max() <= 1e-13 * np.abs(gdata).max()
                nchk += 1
        q.put((rank, nchk, None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_distributed_plan_on_rcm_blocks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rcm, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, n, err in res:
        assert err is None, err
    assert sum(n for _, n, _ in res) == 6 * 80
