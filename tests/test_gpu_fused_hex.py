"""GPU: the fused evaluate-and-assemble kernel for HEXA8 (fs_fused_assemble_hex8) against the two-kernel route it replaces
(fs_solid_batch + fs_csr_numeric_desc, itself pinned to the reference's goldens) and against the oracle directly.
The fused kernel contracts over the Gauss points first (K = C : sum_q g g^T), i.e. it re-associates the same sum of
products: the bar is 1e-12 relative to max|K| against the oracle (the reference's own tolerance class), 1e-13 against the
two-kernel route, EXACT symmetry of the assembled matrix, bitwise repeatability."""
import numpy as np
import pytest
import torch

import fem_shell_amd as fsa
from fem_shell_amd import _lib, batch, meshes
from fem_shell_amd.elements import ElementFamily
from fem_shell_amd.properties import solid_row
from tests import util

pytestmark = pytest.mark.gpu


def brick_block(nx, ny, nz, jitter=0.1, seed=5, shear=0.0):
    ids = np.arange((nx + 1) * (ny + 1) * (nz + 1), dtype=np.int64).reshape(nz + 1, ny + 1, nx + 1)
    Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
    coords = np.stack([X + shear * Y, Y + 0.5 * shear * Z, Z], -1).reshape(-1, 3) * 1e-3
    coords += (meshes.hash_uniform(np.arange(coords.size), seed).reshape(-1, 3) - 0.5) * jitter * 1e-3
    n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
    conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)],
                    -1).reshape(-1, 8)
    return coords, conn


def _asm(coords, conn, mat=None, **kw):
    from fem_shell_amd.assembler import MeshAssembler
    model = {"element_family": ElementFamily.SOLID, "material": mat or util.t300()}
    model.update(kw)
    return MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": model})


def _oracle_csr(oracle, coords, conn, props, pid=None):
    pid = np.zeros(conn.shape[0], np.int32) if pid is None else pid
    Ke = oracle.solid_batch(8, coords[conn], pid, props)["K"]
    dofs = (conn[:, :, None] * 3 + np.arange(3)).reshape(conn.shape[0], -1)
    return oracle.csr_assemble(dofs, coords.shape[0] * 3, Ke)


@pytest.mark.parametrize("dims", [(1, 1, 1), (2, 1, 1), (3, 2, 2), (5, 4, 3), (9, 7, 5), (13, 4, 11), (17, 9, 8)])
def test_fused_hex8_vs_two_kernels_and_oracle(oracle, dims):
    coords, conn = brick_block(*dims, shear=0.2)
    asm = _asm(coords, conn)
    assert asm.can_fuse
    K2 = asm.assemble_stiffness_matrix(recompute=True, fused=False)
    st = torch.full((conn.shape[0],), -1, dtype=torch.int32, device=asm.device)
    Kf = asm.assemble_stiffness_fused(status=st)
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and not torch.isnan(Kf.data).any()
    scale = float(K2.data.abs().max())
    assert float((Kf.data - K2.data).abs().max()) <= 1e-13 * scale
    ip, ix, kd = _oracle_csr(oracle, coords, conn, solid_row(util.t300(), None)[None])
    assert np.array_equal(ip, Kf.indptr.cpu().numpy()) and np.array_equal(ix, Kf.indices.cpu().numpy())
    err = np.abs(Kf.data.cpu().numpy() - kd).max() / np.abs(kd).max()
    print(dims, "fused HEXA8 vs oracle", err, "vs two-kernel route", float((Kf.data - K2.data).abs().max()) / scale,
          asm.fused_plan().stats())
    assert err <= 1e-12
    A = Kf.to_scipy()
    assert (A != A.T).nnz == 0                                           # exactly symmetric
    again = asm.assemble_stiffness_matrix(recompute=True)               # default route == fused; repeatable
    assert asm._fplan is not None and bool((again.data.view(torch.int64) == Kf.data.view(torch.int64)).all())


def test_fused_hex8_renumbered_nodes_and_two_materials(oracle):
    """Random node numbering and element order (the patches come from the coordinates; the CSR slabs of a patch are no
    longer adjacent: many short copy-out runs), and a property table with two rows (the kernel's per-element C path)."""
    coords, conn = brick_block(7, 6, 5, jitter=0.15, seed=9)
    rng = np.random.default_rng(3)
    perm = rng.permutation(coords.shape[0])
    inv = np.empty_like(perm); inv[perm] = np.arange(perm.shape[0])
    coords, conn = coords[perm], inv[conn][rng.permutation(conn.shape[0])]
    asm = _asm(coords, conn)
    iso = fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7850.0)
    rows = np.stack([solid_row(util.t300(), None), solid_row(iso, None)])
    pid = (np.arange(conn.shape[0]) % 2).astype(np.int32)
    asm._prop_table, asm._prop_id = rows, pid
    asm._props_d = batch.to_device_solid_props(rows)
    asm._groups[0]["pid_d"] = torch.from_numpy(pid).to(asm.device)
    K2 = asm.assemble_stiffness_matrix(recompute=True, fused=False)
    Kf = asm.assemble_stiffness_matrix(recompute=True)
    assert asm._fplan is not None
    assert float((Kf.data - K2.data).abs().max()) <= 1e-13 * float(K2.data.abs().max())
    ip, ix, kd = _oracle_csr(oracle, coords, conn, rows, pid)
    assert np.array_equal(ix, Kf.indices.cpu().numpy())
    assert np.abs(Kf.data.cpu().numpy() - kd).max() <= 1e-12 * np.abs(kd).max()
    A = Kf.to_scipy()
    assert (A != A.T).nnz == 0
    print("renumbered:", asm.fused_plan().stats())


def test_fused_hex8_status_and_fallbacks():
    coords, conn = brick_block(4, 3, 3)
    bad = coords.copy()
    bad[conn[5]] = bad[conn[5]][[1, 0, 3, 2, 5, 4, 7, 6]]                 # mirrored brick: det J < 0
    asm = _asm(coords, conn)
    asm._groups[0]["xyz_d"] = torch.from_numpy(bad[conn]).to(asm.device)
    asm.assemble_stiffness_matrix(recompute=True)
    with pytest.raises(ValueError, match="degenerate"):
        asm.check_status()
    # the 1-point rule is not fused: the default entry point keeps the two-kernel route
    red = _asm(coords, conn, reduced_integration=True)
    assert not red.can_fuse
    K = red.assemble_stiffness_matrix(recompute=True)
    assert red._fplan is None and torch.isfinite(K.data).all()
    # a null argument comes back as a code, not as a fault
    lib = _lib.load()
    assert lib.fs_fused_assemble_hex8(None, None, None, 1, 5, 1, None, None, None, None, None, None, None, None, None, None,
                                      _lib.stream_ptr()) == -22


def test_hex_plan_check_rejects_malformed_plans():
    """fs_fused_hex_plan_check: the built plan passes, corrupted copies come back as FS_EINVAL (never launched)."""
    coords, conn = brick_block(9, 8, 7)
    asm = _asm(coords, conn)
    fp = asm.fused_plan()
    lib, dev, nnz, E = _lib.load(), asm.device, asm._plan.nnz, conn.shape[0]
    names = ("pat_eptr", "pat_elems", "pat_nptr", "nodes", "pat_rptr", "runs", "pat_lptr", "steps")

    def check(**repl):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
        a = [t(repl.get(k, getattr(fp, k))) for k in names]
        rc = lib.fs_fused_hex_plan_check(fp.n_patches, *[_lib.ptr(x) for x in a], E, nnz, _lib.stream_ptr())
        return rc, lib.fs_last_error().decode()

    assert check()[0] == 0
    el = fp.pat_elems.copy(); el[3] = E
    rc, msg = check(pat_elems=el); assert rc == -22 and "element index" in msg
    nd = fp.nodes.copy(); nd["sbase"][1] = 19300
    rc, msg = check(nodes=nd); assert rc == -22 and "slab" in msg
    rn = fp.runs.copy(); rn["gdst"][0] = nnz - 3
    rc, msg = check(runs=rn); assert rc == -22 and "run" in msg
    st = fp.steps.copy(); act = np.argwhere((st >> 13) & 1); st[act[5][0], act[5][1]] |= 127
    rc, msg = check(steps=st); assert rc == -22 and "slot" in msg
    st = fp.steps.copy(); fl = np.argwhere((st >> 14) & 1); st[fl[2][0], fl[2][1]] |= np.uint32(127 << 21)
    rc, msg = check(steps=st); assert rc == -22 and "rank" in msg
    lp = fp.pat_lptr.copy(); lp[1:] += 1
    assert check(pat_lptr=lp)[0] == -22


def test_device_hex_partition_on_structured_and_renumbered_blocks(oracle):
    """fused.partition_nodes_hex_device: 4 x 4 x 3-node cells on the lattice of the node numbering (whatever the
    jitter), within the plan's limits; None for a renumbered mesh (the coordinate partition takes over); the matrix
    does not depend on the partition beyond rounding."""
    from fem_shell_amd import fused
    dev = torch.device("cuda", 0)
    coords, conn = brick_block(13, 9, 7, jitter=0.25)
    p = fused.partition_nodes_hex_device(coords.shape[0], torch.from_numpy(conn).to(dev))
    assert p is not None and p.dtype == np.int64 and p.shape == (coords.shape[0],)
    assert np.bincount(p).max() == 48                                          # 4 x 4 x 3 nodes, whatever the jitter
    touched = {(int(q), e) for e in range(conn.shape[0]) for q in p[conn[e]]}
    assert np.bincount([q for q, _ in touched]).max() <= 100
    perm = np.random.default_rng(3).permutation(coords.shape[0])
    assert fused.partition_nodes_hex_device(coords.shape[0], torch.from_numpy(perm[conn]).to(dev)) is None
    asm = _asm(coords, conn)
    K = asm.assemble_stiffness_matrix(recompute=True)
    assert asm._fplan is not None and asm._fplan.nodes.shape[0] / asm._fplan.n_patches > 30
    plan_np = fused.FusedHexPlan(asm._plan.host() if hasattr(asm._plan, "host") else asm._plan, conn, coords)   # NumPy partition
    ref = asm.assemble_stiffness_matrix(recompute=True, fused=False)
    scale = float(ref.data.abs().max())
    assert float((K.data - ref.data).abs().max()) <= 1e-13 * scale
    assert plan_np.n_patches >= asm._fplan.n_patches


@pytest.mark.parametrize("dims", [(7, 5, 4), (13, 9, 7), (33, 17, 9)])
def test_fused_hex8_mass_vs_two_kernels_and_oracle(oracle, dims):
    """fs_fused_assemble_hex8_mass: the consistent mass matrix through the fused plan against element kernel + assembly
    kernel (<= 1e-13 of the largest entry), exactly symmetric, total mass = rho * volume, oracle on the small block."""
    coords, conn = brick_block(*dims, jitter=0.2)
    asm = _asm(coords, conn)
    M2 = asm.assemble_mass_matrix(recompute=True, fused=False)
    Mf = asm.assemble_mass_matrix(recompute=True)
    assert asm._fplan is not None
    scale = float(M2.data.abs().max())
    assert float((Mf.data - M2.data).abs().max()) <= 1e-13 * scale
    A = Mf.to_scipy()
    assert (A != A.T).nnz == 0
    assert bool((asm.assemble_mass_fused().data == Mf.data).all())               # repeatable
    if dims[0] == 7:
        props = solid_row(util.t300(), None)[None]
        Me = oracle.solid_batch(8, coords[conn], np.zeros(conn.shape[0], np.int32), props, want_K=False, want_M=True)["M"]
        dofs = (conn[:, :, None] * 3 + np.arange(3)).reshape(conn.shape[0], -1)
        _, _, md = oracle.csr_assemble(dofs, coords.shape[0] * 3, Me)
        assert np.abs(Mf.data.cpu().numpy() - md).max() <= 1e-12 * np.abs(md).max()
