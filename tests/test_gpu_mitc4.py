"""GPU parity: HIP MITC4+ kernels (through the C ABI) vs the CPU oracle on seeded inputs."""
import numpy as np
import pytest

import fem_shell_amd as fsa
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-12  # north_star: relative 1e-12, measured against max|Ke| per element


def _rel(a, b):
    scale = np.abs(b).reshape(b.shape[0], -1).max(axis=1)
    return (np.abs(a - b).reshape(b.shape[0], -1).max(axis=1) / scale).max()


@pytest.fixture(scope="module")
def setup():
    import torch
    from fem_shell_amd import batch
    props = fsa.build_prop_table(util.standard_props())
    xyz = util.random_quads(1003, seed=1)
    pid = (np.arange(xyz.shape[0]) % len(props)).astype(np.int32)
    dev = torch.device("cuda:0")
    return dict(props=props, xyz=xyz, pid=pid, pd=batch.to_device_props(props),
                xyz_d=torch.from_numpy(xyz).to(dev), pid_d=torch.from_numpy(pid).to(dev))


def test_geometry_bitwise(setup, oracle):
    from fem_shell_amd import batch
    frame, xl, detJ, ac = [t.cpu().numpy() for t in batch.quad_geom(setup["xyz_d"])]
    for e in range(setup["xyz"].shape[0]):
        oe, oxl, odj, oac, _ = oracle.mitc4_geom(setup["xyz"][e])
        assert np.array_equal(frame[e], oe), e
        assert np.array_equal(xl[e], oxl), e
        assert np.array_equal(detJ[e], odj), e
        assert np.abs(ac[e] - oac).max() <= 1e-13 * max(1.0, np.abs(oac).max()), e


def test_K_vs_oracle(setup, oracle):
    from fem_shell_amd import batch
    out = batch.quad_batch(setup["xyz_d"], setup["pid_d"], setup["pd"], len(setup["props"]), dense=True)
    K = out["K"].cpu().numpy()
    Ko, st = oracle.mitc4_K_batch(setup["xyz"], setup["pid"], setup["props"])
    assert np.array_equal(out["status"].cpu().numpy(), st)
    err = _rel(K, Ko)
    print("max rel err K", err)
    assert err < TOL
    assert np.array_equal(K, K.transpose(0, 2, 1))  # exactly symmetric like 0.5*(K+K.T)


def test_M_vs_oracle(setup, oracle):
    from fem_shell_amd import batch
    out = batch.quad_batch(setup["xyz_d"], setup["pid_d"], setup["pd"], len(setup["props"]),
                           want_K=False, want_M=True, dense=True)
    M = out["M"].cpu().numpy()
    Mo, _ = oracle.mitc4_M_batch(setup["xyz"], setup["pid"], setup["props"])
    err = _rel(M, Mo)
    print("max rel err M", err)
    assert err < TOL


def test_ragged_and_empty(setup, oracle):
    import torch
    from fem_shell_amd import batch
    for n in (0, 1, 15, 16, 17):
        out = batch.quad_batch(setup["xyz_d"][:n], setup["pid_d"][:n], setup["pd"], len(setup["props"]), dense=True)
        assert out["K"].shape == (n, 24, 24)
        if n:
            Ko, _ = oracle.mitc4_K_batch(setup["xyz"][:n], setup["pid"][:n], setup["props"])
            assert _rel(out["K"].cpu().numpy(), Ko) < TOL
    with pytest.raises(ValueError):
        bad = torch.full((4,), 99, dtype=torch.int32, device="cuda:0")
        batch.quad_batch(setup["xyz_d"][:4], bad, setup["pd"], len(setup["props"]))


def test_K_M_vs_reference_golden():
    """HIP kernels vs matrices produced by the unmodified reference (tests/golden/elem_mitc4.npz,
    incl. far-from-origin and L/h ~ 3e3 elements, iso + symmetric/asymmetric laminates)."""
    import torch
    from fem_shell_amd import batch
    from tests.util_golden import golden, golden_props
    g = golden("elem_mitc4")
    props = golden_props(g)
    n = g["xyz"].shape[0]
    dev = torch.device("cuda:0")
    out = batch.quad_batch(torch.from_numpy(g["xyz"]).to(dev), torch.arange(n, dtype=torch.int32, device=dev),
                           batch.to_device_props(props), n, want_K=True, want_M=True, dense=True)
    assert not out["status"].any()
    eK, eM = _rel(out["K"].cpu().numpy(), g["Ke"]), _rel(out["M"].cpu().numpy(), g["Me"])
    print("HIP vs reference: K", eK, "M", eM)
    assert eK < TOL and eM < TOL
    fr, xl, dj, ac = [t.cpu().numpy() for t in batch.quad_geom(torch.from_numpy(g["xyz"]).to(dev))]
    assert np.array_equal(fr, g["frame"]) and np.array_equal(xl, g["xl"]) and np.array_equal(dj, g["detJ"])


def test_nonlinear_KT_fint_vs_reference_golden_and_oracle(oracle):
    """update_configuration + compute_tangent_stiffness + compute_internal_forces
    (MITC4.py:1067-1341, MITC4_composite.py:514-621) through fs_shell_quad_batch(KT|FINT)."""
    import torch
    from fem_shell_amd import batch
    from tests.util_golden import golden, golden_props
    g = golden("elem_nl")
    props = golden_props(g)
    n = g["xyz"].shape[0]
    dev = torch.device("cuda:0")
    from fem_shell_amd import _lib
    for layout in (_lib.FS_LAYOUT_SYMBLK, _lib.FS_LAYOUT_SYMBLK_T16):      # per-element and tiled NL kernels
        out = batch.quad_batch(torch.from_numpy(g["xyz"]).to(dev), torch.arange(n, dtype=torch.int32, device=dev),
                               batch.to_device_props(props), n, want_K=False, dense=True,
                               u_e=torch.from_numpy(g["u"]).to(dev), want_KT=True, want_F=True, layout=layout)
        KT, F = out["KT"].cpu().numpy(), out["f_int"].cpu().numpy()
        eK = _rel(KT, g["KT"])
        eF = (np.abs(F - g["f_int"]).max(axis=1) / np.abs(g["f_int"]).max(axis=1)).max()
        print("HIP vs reference, layout", layout, ": KT", eK, "f_int", eF)
        assert eK < 1e-12 and eF < 1e-11
        only_f = batch.quad_batch(torch.from_numpy(g["xyz"]).to(dev), torch.arange(n, dtype=torch.int32, device=dev),
                                  batch.to_device_props(props), n, want_K=False, dense=True,
                                  u_e=torch.from_numpy(g["u"]).to(dev), want_KT=False, want_F=True, layout=layout)
        assert torch.equal(only_f["f_int"], out["f_int"])
    # larger seeded batch vs the oracle (mixed linear / nonlinear rows, iso + laminates)
    ptab = fsa.build_prop_table(util.standard_props(), nonlinear=True)
    ptab2 = fsa.build_prop_table(util.standard_props(), nonlinear=False)
    table = np.concatenate([ptab, ptab2])
    xyz = util.random_quads(300, seed=7, far=False)
    pid = (np.arange(300) % len(table)).astype(np.int32)
    rng = np.random.default_rng(3)
    size = np.linalg.norm(xyz[:, 1] - xyz[:, 0], axis=1)
    u = rng.standard_normal((300, 24)) * 1e-3
    u[:, 0::6] *= size[:, None]; u[:, 1::6] *= size[:, None]; u[:, 2::6] *= size[:, None]
    out = batch.quad_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(pid).to(dev),
                           batch.to_device_props(table), len(table), want_K=False, dense=True,
                           u_e=torch.from_numpy(u).to(dev), want_KT=True, want_F=True)
    KT, F = out["KT"].cpu().numpy(), out["f_int"].cpu().numpy()
    worstK = worstF = 0.0
    for e in range(300):
        ko, fo, _ = oracle.mitc4_NL(xyz[e], table[pid[e]:pid[e] + 1], u[e])
        worstK = max(worstK, np.abs(KT[e] - ko).max() / np.abs(ko).max())
        worstF = max(worstF, np.abs(F[e] - fo).max() / np.abs(fo).max())
    print("HIP vs oracle: KT", worstK, "f_int", worstF)
    assert worstK < 1e-12 and worstF < 1e-11


def test_tiled_layout_equals_element_major(setup):
    """FS_LAYOUT_SYMBLK_T16 (LDS-staged whole-line stores) carries bit for bit the same blocks as
    FS_LAYOUT_SYMBLK: K, M, K_T, f_int on 1003 elements (partial last tile)."""
    import torch
    from fem_shell_amd import _lib, batch
    props = fsa.build_prop_table(util.standard_props(), nonlinear=True)
    pd = batch.to_device_props(props)
    rng = np.random.default_rng(4)
    u = torch.from_numpy(rng.standard_normal((setup["xyz"].shape[0], 24)) * 1e-3).to("cuda:0")
    kw = dict(want_K=True, want_M=True, want_KT=True, want_F=True, u_e=u, dense=True)
    a = batch.quad_batch(setup["xyz_d"], setup["pid_d"], pd, len(props), **kw)
    b = batch.quad_batch(setup["xyz_d"], setup["pid_d"], pd, len(props), layout=_lib.FS_LAYOUT_SYMBLK_T16, **kw)
    for k in ("K", "M", "status"):
        assert torch.equal(a[k], b[k]), k
    # K_T / f_int of the tiled layout come from mitc4_nl_tiled_kernel (Gauss points split over the four lanes
    # of an element, quad reductions): same formulas, sums associated differently
    for k, tol in (("KT", 1e-13), ("f_int", 1e-11)):
        ref = a[k].reshape(a[k].shape[0], -1)
        err = ((b[k].reshape(ref.shape) - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max()
        assert float(err) < tol, (k, float(err))
    raw = batch.quad_batch(setup["xyz_d"], setup["pid_d"], pd, len(props), layout=_lib.FS_LAYOUT_SYMBLK_T16)["K"]
    assert raw.shape == (-(-1003 // 16) * 160, 36)
    with pytest.raises(ValueError):
        batch.expand_blocks(raw, 4, 6, _lib.FS_LAYOUT_SYMBLK_T16)
