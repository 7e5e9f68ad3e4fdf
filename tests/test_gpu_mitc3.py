"""GPU parity: MITC3+ / MITC3Composite batch (fs_shell_tri_batch) vs the oracle and vs the
reference golden vectors (tests/golden/elem_mitc3.npz).  Tolerance 1e-12 * max|Ke| (north_star)."""
import numpy as np
import pytest
import torch

import fem_shell_amd as fsa
from tests import util
from tests.util_golden import golden, golden_props

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _rel(a, b):
    scale = np.abs(b).reshape(b.shape[0], -1).max(axis=1)
    return (np.abs(a - b).reshape(b.shape[0], -1).max(axis=1) / scale).max()


def _run(xyz, pid, table, **kw):
    from fem_shell_amd import batch
    dev = torch.device("cuda:0")
    out = batch.tri_batch(torch.as_tensor(xyz, device=dev), torch.as_tensor(pid, dtype=torch.int32, device=dev),
                          batch.to_device_props(table), len(table), dense=True, **kw)
    torch.cuda.synchronize()
    return {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}


def test_mitc3_vs_reference_goldens():
    g = golden("elem_mitc3")
    props = golden_props(g)
    pid = np.arange(g["xyz"].shape[0], dtype=np.int32)
    out = _run(g["xyz"], pid, props, want_K=True, want_M=True)
    assert not out["status"].any()
    rk, rm = _rel(out["K"], g["Ke"]), _rel(out["M"], g["Me"])
    print("HIP MITC3 vs reference K", rk, "M", rm)
    assert rk < TOL and rm < TOL
    assert np.array_equal(out["K"], np.transpose(out["K"], (0, 2, 1)))   # exactly symmetric


def test_mitc3_vs_oracle_random(oracle):
    rng = np.random.default_rng(5)
    E = 3001    # ragged: not a multiple of the workgroup size
    base = np.array([[0, 0, 0], [1, 0, .05], [.3, 1, .1]], float)
    xyz = base[None] + 0.1 * rng.standard_normal((E, 3, 3))
    xyz = xyz * rng.choice([1.0, 1e-3, 20.0], size=(E, 1, 1)) + rng.choice([0.0, 1e3], size=(E, 1, 1)) * rng.standard_normal((E, 1, 3))
    table = fsa.build_prop_table(util.standard_props())
    pid = rng.integers(0, len(table), E).astype(np.int32)
    K, M, st = oracle.mitc3_batch(xyz, pid, table, want_K=True, want_M=True, nthreads=8)
    out = _run(xyz, pid, table, want_K=True, want_M=True)
    assert np.array_equal(out["status"], st)
    rk, rm = _rel(out["K"], K), _rel(out["M"], M)
    print("HIP MITC3 vs oracle K", rk, "M", rm)
    assert rk < TOL and rm < TOL


def test_mitc3_degenerate_and_empty(oracle):
    table = fsa.build_prop_table(util.standard_props())
    xyz = np.array([[[0, 0, 0], [1, 0, 0], [2, 0, 0]],          # collinear -> ValueError in the reference
                    [[0, 0, 0], [1, 0, 0], [0, 1, 0]]], float)
    out = _run(xyz, np.zeros(2, np.int32), table)
    assert out["status"][0] == 1 and out["status"][1] == 0
    out = _run(np.zeros((0, 3, 3)), np.zeros(0, np.int32), table)
    assert out["K"].shape == (0, 18, 18)


def test_wave_kernel_ragged_batch_sizes(oracle):
    """mitc3_k_lds_kernel owns 10 triangles per wave: 32 batch sizes with ragged last waves against the oracle."""
    table = fsa.build_prop_table(util.standard_props())
    rng = np.random.default_rng(3)
    worst = 0.0
    for E in list(range(1, 26)) + [59, 60, 61, 639, 640, 641, 10007]:
        xyz = rng.standard_normal((E, 3, 3)) + rng.uniform(-50, 50, (E, 1, 3))
        pid = rng.integers(0, len(table), E).astype(np.int32)
        K, _, st = oracle.mitc3_batch(xyz, pid, table, want_K=True, want_M=False, nthreads=8)
        out = _run(xyz, pid, table, want_K=True)
        assert np.array_equal(out["status"], st), E
        ok = st == 0
        if ok.any():
            worst = max(worst, _rel(out["K"][ok], K[ok]))
    print("ragged MITC3 batches vs oracle", worst)
    assert worst < TOL
