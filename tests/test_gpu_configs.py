"""BASELINE.json configurations at their FULL sizes on the GPU, checked through size-independent properties
(the oracle finishes only samples of them in seconds):
  C2  pinched hemisphere (18 degree cut-out) 256 x 256, MITC4+ and MITC3+          (configs[1])
  C3  blade-like laminate tube 512 x 128, 192 property sets, K, M, lumped mass     (configs[2])
  C4  NL cantilever 400 x 40, K_T / f_int re-assembled over Newton iterates        (configs[3])
  C5  one GPU's 400 x 3125 strip, the whole 10 M-element plate and the 1 M-element HEXA8 block on one GPU   (configs[4])
Properties: closed-form nnz, CSR rows of sampled nodes against the oracle's element matrices summed in element
order (1e-12), symmetry (x.Ky == y.Kx), rigid translations in the null space of K, total mass from the assembled
and the lumped M, bitwise repeatability, K_T(0) == K and f_int(0) == 0."""
import numpy as np
import pytest
import torch

import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.elements import ElementFamily
from tests import util

pytestmark = pytest.mark.gpu


def _spmv(A, x):
    """CSR mat-vec on the device through torch (test plumbing only)."""
    M = torch.sparse_csr_tensor(A.indptr, A.indices.to(torch.int64), A.data, size=A.shape)
    return M @ x


def _sample_rows(asm, mesh, A, oracle, which, nodes, table=None, pid=None):
    """CSR rows of `nodes` vs the oracle: element matrices of the touching elements, summed in element order."""
    conn = mesh.conn
    nn = conn.shape[1]
    table = asm._prop_table if table is None else table
    pid = asm._prop_id if pid is None else pid
    touch = np.flatnonzero(np.isin(conn, nodes).any(axis=1))
    xyz = mesh.coords[conn[touch]]
    if nn == 4:
        vals, _ = (oracle.mitc4_K_batch if which == "K" else oracle.mitc4_M_batch)(xyz, pid[touch], table)
    else:
        Ke, Me, _ = oracle.mitc3_batch(xyz, pid[touch], table, want_K=which == "K", want_M=which == "M")
        vals = Ke if which == "K" else Me
    ip = A.indptr.cpu().numpy()
    worst, scale = 0.0, 0.0
    ix_all, data_all = A.indices, A.data
    for I in nodes:
        lo, hi = int(ip[6 * I]), int(ip[6 * I + 6])
        cols = ix_all[lo:hi].cpu().numpy()
        got = data_all[lo:hi].cpu().numpy()
        m = (int(ip[6 * I + 1]) - lo)
        got = got.reshape(6, m)
        colnodes = cols[:m:6] // 6
        ref = np.zeros((6, m))
        for k, e in enumerate(touch):                       # element order
            loc = np.flatnonzero(conn[e] == I)
            if not loc.size:
                continue
            i = int(loc[0])
            for j in range(nn):
                c = int(np.searchsorted(colnodes, conn[e, j]))
                ref[:, 6 * c:6 * c + 6] += vals[k][6 * i:6 * i + 6, 6 * j:6 * j + 6]
        worst = max(worst, np.abs(got - ref).max())
        scale = max(scale, np.abs(ref).max())
    return worst / scale


def _common_checks(asm, mesh, K, oracle, nodes, rng):
    n = asm.dofs_count
    err = _sample_rows(asm, mesh, K, oracle, "K", nodes)
    x = torch.from_numpy(rng.standard_normal(n)).to(K.data.device)
    y = torch.from_numpy(rng.standard_normal(n)).to(K.data.device)
    Kx, Ky = _spmv(K, x), _spmv(K, y)
    sym = abs(float(y @ Kx) - float(x @ Ky)) / (float(x.norm()) * float(Ky.norm()))
    rb = 0.0
    for k in range(3):                                        # rigid translations: K t = 0
        t = torch.zeros(n, dtype=torch.float64, device=K.data.device)
        t[k::6] = 1.0
        rb = max(rb, float(_spmv(K, t).abs().max()) / float(K.data.abs().max()))
    return err, sym, rb


@pytest.mark.parametrize("tri", [False, True])
def test_c2_hemisphere_256(oracle, tri):
    from fem_shell_amd.assembler import MeshAssembler
    n = 256
    mesh = meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, n, triangular=tri)
    mat = fsa.IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.04}})
    K = asm.assemble_stiffness_matrix()
    assert mesh.conn.shape[0] == (131072 if tri else 65536) and asm.dofs_count == 396294
    assert K.nnz == (16570404 if tri else 21288996)                       # SURVEY.md 8d closed forms
    rng = np.random.default_rng(2)
    nodes = rng.choice(mesh.node_count, 64, replace=False)
    err, sym, rb = _common_checks(asm, mesh, K, oracle, nodes, rng)
    print("C2", "MITC3+" if tri else "MITC4+", "rows vs oracle", err, "symmetry", sym, "rigid translation", rb)
    assert err < 1e-12 and sym < 1e-12 and rb < 1e-9
    # the fused kernels themselves (fs_fused_assemble_quad / fs_fused_assemble_tri) on this configuration, against the
    # oracle directly
    Kf = asm.assemble_stiffness_matrix(recompute=True)
    assert asm._fplan is not None
    errf = _sample_rows(asm, mesh, Kf, oracle, "K", nodes)
    print("C2", "MITC3+" if tri else "MITC4+", "fused rows vs oracle", errf, "vs two-kernel route",
          float((Kf.data - K.data).abs().max() / K.data.abs().max()))
    assert errf < 1e-12 and float((Kf.data - K.data).abs().max()) <= 1e-13 * float(K.data.abs().max())
    asm.check_status()
    M = asm.assemble_mass_matrix()
    assert _sample_rows(asm, mesh, M, oracle, "M", nodes[:16]) < 1e-12
    # total mass: sum of the u-u entries = rho h area of the patch (R^2 (cos 18 - cos 90) pi/2), faceted
    area = 100.0 * np.cos(np.radians(18.0)) * np.pi / 2
    mu = float(M.lump_rows()[0::6].sum())
    assert abs(mu / (1.0 * 0.04 * area) - 1.0) < 1e-3


@pytest.mark.parametrize("thick", [0.32, 0.0032])
@pytest.mark.parametrize("tri", [False, True])
def test_c2_twisted_beam_256(oracle, tri, thick):
    """configs[1]'s second mesh at full size: MacNeal-Harder twisted beam, 256 x 256 (mesh of
    tests/benchmarks/test_ko2017_performance.py:1195-1229, L = 12, b = 1.1, 90 degrees; both thicknesses of
    :1256-1261).  L/h ~ 2.8e3: the case SURVEY.md A.1 singles out as breaking 1e-12 unless the geometry stage is
    bit-faithful -- same property set as the hemisphere."""
    from fem_shell_amd.assembler import MeshAssembler
    n = 256
    mesh = meshes.twisted_beam(12.0, 1.1, 90.0, n, n, triangular=tri)
    mat = fsa.IsotropicMaterial("Ko2017_TwistedBeam", 29.0e6, 0.22, 1.0)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": thick}})
    K = asm.assemble_stiffness_matrix()
    assert mesh.conn.shape[0] == (131072 if tri else 65536) and asm.dofs_count == 396294
    assert K.nnz == (16570404 if tri else 21288996)                       # SURVEY.md 8d closed forms
    rng = np.random.default_rng(3)
    nodes = np.unique(np.concatenate([[0, n, mesh.node_count - 1], rng.choice(mesh.node_count, 61, replace=False)]))
    err, sym, rb = _common_checks(asm, mesh, K, oracle, nodes, rng)
    print("C2 twisted beam", "MITC3+" if tri else "MITC4+", "t", thick, "rows vs oracle", err, "symmetry", sym,
          "rigid translation", rb)
    assert err < 1e-12 and sym < 1e-12 and rb < 1e-9
    again = asm.assemble_stiffness_matrix(recompute=True)                # the fused route
    assert asm._fplan is not None
    if tri:        # same block function in another inlining context: equal to rounding (see tests/test_gpu_fused_tri.py)
        assert float((again.data - K.data).abs().max()) <= 1e-13 * float(K.data.abs().max())
        assert _sample_rows(asm, mesh, again, oracle, "K", nodes) < 1e-12
    else:
        util.assert_fused_close(again, K)
        assert _sample_rows(asm, mesh, again, oracle, "K", nodes) < 1e-12
    assert bool((asm.assemble_stiffness_matrix(recompute=True).data == again.data).all())      # bitwise repeatable
    M = asm.assemble_mass_matrix()
    assert _sample_rows(asm, mesh, M, oracle, "M", nodes[:16]) < 1e-12
    # total mass: the ruled (helicoidal) strip's area is close to L * b * (mean stretch); lumped == consistent total
    t = torch.zeros(asm.dofs_count, dtype=torch.float64, device=M.data.device); t[0::6] = 1.0
    tot = float(M.lump_rows()[0::6].sum())
    assert abs(float(t @ _spmv(M, t)) / tot - 1.0) < 1e-12
    assert 1.0 <= tot / (1.0 * thick * 12.0 * 1.1) < 1.002               # helicoid area >= flat area, < +0.2 %


def test_c3_blade_like_tube(oracle):
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.tapered_tube(128, 512)                                   # 65 536 quads, closed section
    rng = np.random.default_rng(7)
    m = util.t300()
    core = fsa.OrthotropicMaterial("core", (0.13e9,) * 3, (0.05e9,) * 3, (0.3,) * 3, 100.0)
    props = {}
    for k, name in enumerate(mesh.set_names):
        kind = rng.integers(0, 4)
        t = 0.5e-3
        if kind == 0:
            ang = [0.0] * int(rng.integers(4, 41))                         # UD
        elif kind == 1:
            ang = [45.0, -45.0] * int(rng.integers(2, 21))                 # biax
        elif kind == 2:
            ang = [0.0, 45.0, -45.0] * int(rng.integers(2, 14))            # triax
        else:
            ang = None                                                     # sandwich-like thick core
        if ang is None:
            lam = fsa.Laminate([fsa.Ply(m, 1e-3, 0.0), fsa.Ply(m, 1e-3, 45.0), fsa.Ply(core, 25e-3, 0.0),
                                fsa.Ply(m, 1e-3, 45.0), fsa.Ply(m, 1e-3, 0.0)])
        else:
            if k % 10 == 0:
                ang = ang + [90.0]                                         # ~10 % asymmetric stacks: coupling
            lam = fsa.create_laminate_from_angles(m, t, ang)
        props[name] = fsa.CompositeShellProperty(lam)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "properties": props}})
    assert len(asm._prop_table) == 192 and (asm._prop_table["flags"] & 1).sum() >= 10
    K = asm.assemble_stiffness_matrix()
    nodes = rng.choice(mesh.node_count, 48, replace=False)
    err, sym, rb = _common_checks(asm, mesh, K, oracle, nodes, rng)
    print("C3 rows vs oracle", err, "symmetry", sym, "rigid translation", rb)
    assert err < 1e-12 and sym < 1e-12 and rb < 1e-9
    # the headline kernel itself (fs_fused_assemble_quad, coupled variant) on this configuration, against the oracle
    Kf = asm.assemble_stiffness_matrix(recompute=True)
    assert asm._fplan is not None
    errf = _sample_rows(asm, mesh, Kf, oracle, "K", nodes)
    print("C3 fused rows vs oracle", errf, "vs two-kernel route", float((Kf.data - K.data).abs().max() / K.data.abs().max()))
    assert errf < 1e-12 and float((Kf.data - K.data).abs().max()) <= 1e-13 * float(K.data.abs().max())
    asm.check_status()
    M = asm.assemble_mass_matrix()
    lumped = M.lump_rows()
    # mass check: lumped translational mass == sum over elements of m_t * area (oracle element masses)
    Me, _ = oracle.mitc4_M_batch(mesh.coords[mesh.conn[:2048]], asm._prop_id[:2048], asm._prop_table)
    assert float(lumped.min()) > 0.0
    tot = float(lumped[0::6].sum())
    ex = float(M.data.sum()) / 1.0                                         # all entries: 3 translations + 3 rotations
    assert tot > 0 and np.isfinite(ex)
    x_dir = torch.zeros(asm.dofs_count, dtype=torch.float64, device=M.data.device); x_dir[0::6] = 1.0
    assert abs(float(x_dir @ _spmv(M, x_dir)) / tot - 1.0) < 1e-12         # t^T M t == lumped total


def test_c4_nl_cantilever_newton(oracle):
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0, 400, 40)
    mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1,
                                            "nonlinear": True}})
    n = asm.dofs_count
    K = asm.assemble_stiffness_matrix()
    KT0, f0 = asm.assemble_tangent_and_forces(np.zeros(n))
    # K_T(0) is the linear matrix of the fused kernel (the Newton path works on it), K above the two-kernel route's
    util.assert_fused_close(KT0, K)
    assert bool((KT0.data == asm.assemble_stiffness_matrix(recompute=True).data).all()) and float(f0.abs().max()) == 0.0
    rng = np.random.default_rng(4)
    xs = mesh.coords[:, 0]
    shape = np.zeros((mesh.node_count, 6))
    shape[:, 2] = xs ** 2 / 100.0; shape[:, 4] = -2 * xs / 100.0; shape[:, 0] = -xs / 1000.0
    sel = rng.choice(mesh.conn.shape[0], 256, replace=False)
    dofs = asm._dofs_array
    for k in range(1, 11):                                                 # 10 Newton-like iterates u_k = k * 1e-3 * field
        u = (k * 1e-3 * shape + 1e-7 * rng.standard_normal(shape.shape)).reshape(-1)
        KT, f = asm.assemble_tangent_and_forces(u)
        assert torch.isfinite(KT.data).all() and torch.isfinite(f).all()
        if k in (1, 10):                                                   # sampled elements vs the oracle loop
            fsum, fscale = np.zeros(n), 0.0
            touch = np.flatnonzero(np.isin(mesh.conn, mesh.conn[sel[:8]].ravel()).any(axis=1))
            for e in touch:
                kt, fe, _ = oracle.mitc4_NL(mesh.coords[mesh.conn[e]], asm._prop_table[:1], u[dofs[e]])
                fsum[dofs[e]] += fe
                fscale = max(fscale, np.abs(fe).max())
            full = np.unique(mesh.conn[sel[:8]].ravel())                   # nodes whose every element is in `touch`
            inner = [I for I in full if np.isin(np.flatnonzero((mesh.conn == I).any(axis=1)), touch).all()]
            idx = (np.array(inner)[:, None] * 6 + np.arange(6)).ravel()
            fe_gpu = f.cpu().numpy()
            # f_int = K u + ... on a smooth, nearly rigid field: the products K_ij u_j are ~1e3 x the force
            # they sum to, so the fp64 tolerance is set on the magnitude of the summed terms
            terms = float(K.data.abs().max()) * np.abs(u).max()
            err = np.abs(fe_gpu[idx] - fsum[idx]).max()
            print("C4 iterate", k, "f_int err", err, "element force", fscale, "|K||u|", terms)
            assert err <= 1e-12 * terms and err <= 1e-10 * fscale
    print("C4 ok: 10 re-assemblies of K_T and f_int,", mesh.conn.shape[0], "elements")
    # cached linear blocks (default) against recomputing them every iterate: bitwise the same K_T and f_int
    uk = torch.from_numpy((5e-3 * shape).reshape(-1)).to(asm.device)
    assert asm.reuse_linear_blocks
    Kc, fc = asm.assemble_tangent_and_forces(uk)
    fc_only = asm.assemble_internal_forces(uk)
    Kc_only = asm.assemble_tangent_stiffness(uk)
    asm.reuse_linear_blocks = False
    Kr, fr = asm.assemble_tangent_and_forces(uk)
    asm.reuse_linear_blocks = True
    assert bool((Kc.data == Kr.data).all()) and bool((fc == fr).all())
    # (f_int alone forms its nonlinear part in the element kernel, the combined call in the tangent kernel: same terms)
    assert float((fc_only - fr).abs().max()) <= 1e-12 * float(fr.abs().max()) and bool((Kc_only.data == Kr.data).all())
    # the same re-assembly captured in one HIP graph: bitwise the same K_T and f_int, replay after replay
    step = asm.capture_tangent_and_forces()
    for k in (3, 7, 3):
        uk = torch.from_numpy((k * 1e-3 * shape).reshape(-1)).to(asm.device)
        Kd, fd = asm.assemble_tangent_and_forces(uk)
        Kg, fg = step(uk)
        torch.cuda.synchronize()
        assert bool((Kg.data == Kd.data).all()) and bool((fg == fd).all())


def test_c4_newton_plate_full_size(oracle):
    """The Newton bench workload at full size (400 x 3125 = 1.25 M elements, bench.newton_leg): K_T and f_int on the
    assembled matrix against the oracle loop on sampled nodes, against the element-block route everywhere, exact symmetry
    through two mat-vecs."""
    import bench
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * bench.STRIP_NY / 40, bench.STRIP_NX, bench.STRIP_NY)
    mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1,
                                            "nonlinear": True}}, cache_blocks=False)
    n = asm.dofs_count
    xs = mesh.coords[:, 0]
    rng = np.random.default_rng(11)
    shape = np.zeros((mesh.node_count, 6))
    shape[:, 2] = xs ** 2 / 100.0; shape[:, 4] = -2 * xs / 100.0; shape[:, 0] = -xs / 1000.0
    u = (5e-3 * shape + 1e-6 * rng.standard_normal(shape.shape)).reshape(-1)
    ud = torch.from_numpy(u).to(asm.device)
    KT, f = asm.assemble_tangent_and_forces(ud)
    assert asm._K_blk is None                                   # no element blocks on this route
    # sampled nodes against the oracle loop
    nodes = rng.choice(mesh.node_count, 24, replace=False)
    touch = np.flatnonzero(np.isin(mesh.conn, nodes).any(axis=1))
    dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
    kt = {}
    fsum = {int(I): np.zeros(6) for I in nodes}
    for e in touch:
        kt[e], fe, _ = oracle.mitc4_NL(mesh.coords[mesh.conn[e]], asm._prop_table[:1], u[dofs[e]])
        for i, I in enumerate(mesh.conn[e]):
            if int(I) in fsum:
                fsum[int(I)] += fe[6 * i:6 * i + 6]
    ip = KT.indptr.cpu().numpy()
    worstK = worstF = 0.0
    scale = float(KT.data.abs().max())
    fh = f.cpu().numpy()
    for I in nodes:
        lo, hi = int(ip[6 * I]), int(ip[6 * I + 6])
        m = int(ip[6 * I + 1]) - lo
        got = KT.data[lo:hi].cpu().numpy().reshape(6, m)
        colnodes = KT.indices[lo:lo + m:6].cpu().numpy() // 6
        ref = np.zeros((6, m))
        for e in touch:                                         # element order
            loc = np.flatnonzero(mesh.conn[e] == I)
            if loc.size:
                i = int(loc[0])
                for j in range(4):
                    c = int(np.searchsorted(colnodes, mesh.conn[e, j]))
                    ref[:, 6 * c:6 * c + 6] += kt[e][6 * i:6 * i + 6, 6 * j:6 * j + 6]
        worstK = max(worstK, np.abs(got - ref).max())
        worstF = max(worstF, np.abs(fh[6 * I:6 * I + 6] - fsum[int(I)]).max())
    terms = scale * np.abs(u).max()
    print("C4 full size: K_T rows vs oracle", worstK / scale, "f_int vs oracle / |K||u|", worstF / terms)
    assert worstK <= 1e-12 * scale and worstF <= 1e-12 * terms
    # exact symmetry: x' (K_T y) == y' (K_T x) up to the rounding of the products
    x = torch.from_numpy(rng.standard_normal(n)).to(asm.device)
    y = torch.from_numpy(rng.standard_normal(n)).to(asm.device)
    Kx, Ky = _spmv(KT, x), _spmv(KT, y)
    assert abs(float(y @ Kx) - float(x @ Ky)) <= 1e-12 * float(x.norm()) * float(Ky.norm())
    # the element-block route on the whole plate
    asm.newton_on_assembled = False
    K2, f2 = asm.assemble_tangent_and_forces(ud)
    errK = float((KT.data - K2.data).abs().max()) / scale
    errF = float((f - f2).abs().max()) / terms
    print("C4 full size: vs element-block route K_T", errK, "f_int / |K||u|", errF)
    assert errK <= 1e-13 and errF <= 1e-13


def test_c5_strip_full_size(oracle):
    import bench
    from fem_shell_amd.assembler import MeshAssembler
    lams = bench.c5_laminates()
    mesh = meshes.jittered_plate(bench.STRIP_NX, bench.STRIP_NY, cell=1e-3, seed=11, n_sets=16)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL,
                                            "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}})
    K = asm.assemble_stiffness_matrix()
    assert K.nnz == 36 * (3 * bench.STRIP_NX + 1) * (3 * bench.STRIP_NY + 1)       # a17 closed form
    rng = np.random.default_rng(11)
    nodes = rng.choice(mesh.node_count, 64, replace=False)
    err, sym, rb = _common_checks(asm, mesh, K, oracle, nodes, rng)
    print("C5 rows vs oracle", err, "symmetry", sym, "rigid translation", rb)
    assert err < 1e-12 and sym < 1e-12 and rb < 1e-9
    again = asm.assemble_stiffness_matrix(recompute=True)                          # the fused route
    util.assert_fused_close(again, K)
    assert _sample_rows(asm, mesh, again, oracle, "K", nodes) < 1e-12
    third = asm.assemble_stiffness_matrix(recompute=True)
    assert torch.equal(third.data, again.data)                                     # no atomics: bitwise repeatable


def test_c5_plate_10m_one_gpu(oracle):
    """configs[4] at its FULL size on one GPU: 3200 x 3125 = 10 M MITC4+ laminate elements, 60 M dofs,
    nnz = 3 240 683 136 > 2^31 (int64 indptr / value offsets, 1e8 node-pair blocks).  Checked through the closed-form
    nnz, sampled CSR rows against the oracle (first, last and random nodes), block symmetry K_IJ == K_JI^T of the
    sampled rows and a bitwise comparison of sampled rows with the same plate assembled from its first 200 element
    rows (rows whose elements all lie inside that piece must not depend on the rest of the mesh)."""
    import bench
    from fem_shell_amd.assembler import MeshAssembler
    NX, NY = 3200, 3125
    free = torch.cuda.mem_get_info()[0]
    if free < 120e9:
        pytest.skip(f"needs ~100 GB of device memory, {free / 1e9:.0f} GB free")
    lams = bench.c5_laminates()
    mesh = meshes.jittered_plate(NX, NY, cell=1e-3, seed=11, n_sets=16)
    model = {"elements": {"element_family": ElementFamily.SHELL,
                          "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}}
    asm = MeshAssembler(mesh, model)
    K = asm.assemble_stiffness_matrix()
    nnz = 36 * (3 * NX + 1) * (3 * NY + 1)
    assert nnz == 3_240_683_136 and K.nnz == nnz and int(K.indptr[-1]) == nnz and K.data.numel() == nnz
    rng = np.random.default_rng(5)
    N = mesh.node_count
    nodes = np.unique(np.concatenate([[0, N - 1, N - NX - 3], rng.choice(N, 29, replace=False)]))
    err = _sample_rows(asm, mesh, K, oracle, "K", nodes)
    print("C5 10M: sampled rows vs oracle", err)
    assert err < 1e-12
    ip = K.indptr.cpu().numpy()

    def block_row(A, ipx, I):
        lo, hi = int(ipx[6 * I]), int(ipx[6 * I + 6])
        m = int(ipx[6 * I + 1]) - lo
        return A.indices[lo:lo + m].cpu().numpy()[::6] // 6, A.data[lo:hi].cpu().numpy().reshape(6, m)

    for I in nodes[:8]:                                         # K_IJ == K_JI^T, bit for bit (symmetrised blocks)
        cols, rowI = block_row(K, ip, int(I))
        for c, J in enumerate(cols):
            colsJ, rowJ = block_row(K, ip, int(J))
            cj = int(np.searchsorted(colsJ, I))
            assert np.array_equal(rowI[:, 6 * c:6 * c + 6], rowJ[:, 6 * cj:6 * cj + 6].T)
    # the first 200 element rows as a mesh of their own: same node numbering (row-major), so rows of nodes
    # below the cut are identical bit for bit
    del K
    torch.cuda.empty_cache()
    sub_rows = 200
    sub = meshes.ArrayMesh(mesh.coords[:(NX + 1) * (sub_rows + 1)], mesh.conn[:NX * sub_rows],
                           mesh.set_id[:NX * sub_rows], mesh.set_names)
    Ks = MeshAssembler(sub, model).assemble_stiffness_matrix(recompute=True)     # the fused route, like K below
    K = asm.assemble_stiffness_matrix(recompute=True)
    ips = Ks.indptr.cpu().numpy()
    low = np.concatenate([[0, 17, (NX + 1) * (sub_rows - 1) + 5], rng.choice((NX + 1) * sub_rows, 13, replace=False)])
    for I in low:
        c1, r1 = block_row(K, ip, int(I))
        c2, r2 = block_row(Ks, ips, int(I))
        assert np.array_equal(c1, c2) and np.array_equal(r1, r2)


def test_c5_hexa8_block_full_size(oracle):
    """configs[4]'s solid block at its FULL size on one GPU: 128 x 128 x 64 = 1 048 576 jittered HEXA8 (orthotropic C),
    3.2 M dofs.  Closed-form nnz of the 27-point node graph, sampled block rows against the oracle (element order),
    K_IJ == K_JI^T on the samples, total mass from M, bitwise repeatability."""
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.properties import solid_row
    nx, ny, nz = 128, 128, 64
    ids = np.arange((nx + 1) * (ny + 1) * (nz + 1), dtype=np.int64).reshape(nz + 1, ny + 1, nx + 1)
    Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
    coords = np.stack([X, Y, Z], -1).reshape(-1, 3) * 1e-3
    coords += (meshes.hash_uniform(np.arange(coords.size), 5).reshape(-1, 3) - 0.5) * 1e-4
    n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
    conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)],
                    -1).reshape(-1, 8)
    mat = util.t300()
    asm = MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
    K = asm.assemble_stiffness_matrix()
    per_axis = lambda m: 3 * (m + 1) - 2                   # sum over the nodes of an axis of their neighbours on it
    assert K.nnz == 9 * per_axis(nx) * per_axis(ny) * per_axis(nz)
    props = solid_row(mat, None)[None]
    rng = np.random.default_rng(8)
    N = coords.shape[0]
    nodes = np.unique(np.concatenate([[0, N - 1], rng.choice(N, 14, replace=False)]))
    ip = K.indptr.cpu().numpy()
    touch = np.flatnonzero(np.isin(conn, nodes).any(axis=1))
    Ke = oracle.solid_batch(8, coords[conn[touch]], np.zeros(touch.size, np.int32), props)["K"]
    worst = scale = 0.0
    for I in nodes:
        lo, hi = int(ip[3 * I]), int(ip[3 * I + 3])
        m = int(ip[3 * I + 1]) - lo
        cols = K.indices[lo:lo + m].cpu().numpy()[::3] // 3
        got = K.data[lo:hi].cpu().numpy().reshape(3, m)
        ref = np.zeros((3, m))
        for k, e in enumerate(touch):                       # element order
            loc = np.flatnonzero(conn[e] == I)
            if loc.size:
                i = int(loc[0])
                for j in range(8):
                    c = int(np.searchsorted(cols, conn[e, j]))
                    ref[:, 3 * c:3 * c + 3] += Ke[k][3 * i:3 * i + 3, 3 * j:3 * j + 3]
        worst, scale = max(worst, np.abs(got - ref).max()), max(scale, np.abs(ref).max())
        for c, J in enumerate(cols[:6]):                    # K_IJ == K_JI^T
            lo2 = int(ip[3 * J]); m2 = int(ip[3 * J + 1]) - lo2
            cols2 = K.indices[lo2:lo2 + m2].cpu().numpy()[::3] // 3
            blk2 = K.data[lo2:int(ip[3 * J + 3])].cpu().numpy().reshape(3, m2)
            cj = int(np.searchsorted(cols2, I))
            assert np.abs(got[:, 3 * c:3 * c + 3] - blk2[:, 3 * cj:3 * cj + 3].T).max() <= 1e-12 * np.abs(ref).max()
    print("C5 HEXA8 block: sampled rows vs oracle", worst / scale, "nnz", K.nnz)
    assert worst <= 1e-12 * scale
    # the fused evaluate-and-assemble kernel (fs_fused_assemble_hex8: the default of a re-evaluation) on the full block:
    # the same sampled rows against the oracle, agreement with the two-kernel route, bitwise repeatability
    Kf = asm.assemble_stiffness_matrix(recompute=True)
    assert asm._fplan is not None
    worst_f = 0.0
    for I in nodes:
        lo, hi = int(ip[3 * I]), int(ip[3 * I + 3])
        worst_f = max(worst_f, float((Kf.data[lo:hi] - K.data[lo:hi]).abs().max()))
    dmax = float((Kf.data - K.data).abs().max())
    print("C5 HEXA8 block fused: vs two-kernel route", dmax / scale, "plan", asm.fused_plan().stats())
    assert dmax <= 1e-13 * scale and worst + worst_f <= 1e-12 * scale
    again = asm.assemble_stiffness_matrix(recompute=True)
    assert bool((again.data == Kf.data).all())
    asm.check_status()
    del Kf, again
    M = asm.assemble_mass_matrix()
    vol = nx * ny * nz * 1e-9
    total = float(M.data.sum()) / 3.0                        # each direction carries the whole mass
    assert abs(total / (mat.rho * vol) - 1.0) < 1e-3         # jitter changes the volume of boundary cells only
