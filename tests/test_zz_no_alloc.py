"""GPU: fs_shell_quad_fint_noalloc (the one entry point that allocates nothing; include/femshell_hip.h) against the
existing entry point it is a variant of, fs_shell_quad_batch(FS_WANT_FINT) with a caller-provided KTe workspace.
Both run the same kernels on the same inputs, so the bar is bit-exact.  Two calls in a row, the second one on a
smaller input (the static workspace keeps the first call's contents), plus the oracle as an independent check and
the size limit."""
import ctypes

import numpy as np
import pytest
import torch

from fem_shell_amd import _lib
from fem_shell_amd.batch import to_device_props
from fem_shell_amd.properties import build_prop_table
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _inputs(E, seed):
    rng = np.random.default_rng(seed)
    props = build_prop_table(util.standard_props(), nonlinear=True)
    xyz = util.random_quads(E, seed=seed, far=False)
    pid = rng.integers(0, props.shape[0], E).astype(np.int32)
    u = 1e-3 * rng.standard_normal((E, 24))
    return props, xyz, pid, u


def _existing(lib, xyz_d, pid_d, props_d, n_props, u_d, E):
    """fs_shell_quad_batch, FS_WANT_FINT, tiled layout, caller-provided KTe workspace."""
    ws = torch.empty((-(-E // 16) * 160, 36), dtype=torch.float64, device=xyz_d.device)
    fe = torch.full((E, 24), np.nan, dtype=torch.float64, device=xyz_d.device)
    st = torch.zeros(E, dtype=torch.int32, device=xyz_d.device)
    _lib.check(lib.fs_shell_quad_batch(_lib.ptr(xyz_d), _lib.ptr(pid_d), _lib.ptr(props_d), n_props, _lib.ptr(u_d), E,
                                       _lib.FS_WANT_FINT, _lib.FS_LAYOUT_SYMBLK_T16, None, None, _lib.ptr(ws),
                                       _lib.ptr(fe), _lib.ptr(st), _lib.stream_ptr()), "fs_shell_quad_batch")
    return fe, st


def _noalloc(lib, xyz_d, pid_d, props_d, n_props, u_d, E):
    fe = torch.full((E, 24), np.nan, dtype=torch.float64, device=xyz_d.device)
    st = torch.zeros(E, dtype=torch.int32, device=xyz_d.device)
    _lib.check(lib.fs_shell_quad_fint_noalloc(_lib.ptr(xyz_d), _lib.ptr(pid_d), _lib.ptr(props_d), n_props,
                                              _lib.ptr(u_d), E, _lib.ptr(fe), _lib.ptr(st), _lib.stream_ptr()),
               "fs_shell_quad_fint_noalloc")
    return fe, st


def test_noalloc_fint_matches_existing_entry_point_twice():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    for E, seed in ((3001, 5), (517, 6)):          # second call: smaller input, ragged last tile, stale workspace
        props, xyz, pid, u = _inputs(E, seed)
        props_d = to_device_props(props)
        xyz_d, pid_d, u_d = (torch.from_numpy(a).to(dev) for a in (xyz, pid, u))
        ref, st_ref = _existing(lib, xyz_d, pid_d, props_d, props.shape[0], u_d, E)
        got, st = _noalloc(lib, xyz_d, pid_d, props_d, props.shape[0], u_d, E)
        torch.cuda.synchronize()
        assert torch.equal(st, st_ref)
        assert not torch.isnan(got).any()
        assert torch.equal(got, ref), float((got - ref).abs().max())          # same kernels: bit for bit
        # independent check against the CPU oracle on a slice
        n = 64
        f_or = np.stack([O.mitc4_NL(xyz[e], props[pid[e]], u[e])[1] for e in range(n)])
        g = got[:n].cpu().numpy()
        assert np.abs(g - f_or).max() <= 1e-11 * np.abs(f_or).max()


def test_noalloc_fint_limits():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    props, xyz, pid, u = _inputs(16, 1)
    props_d = to_device_props(props)
    xyz_d, pid_d, u_d = (torch.from_numpy(a).to(dev) for a in (xyz, pid, u))
    fe = torch.empty((16, 24), dtype=torch.float64, device=dev)
    args = (_lib.ptr(xyz_d), _lib.ptr(pid_d), _lib.ptr(props_d), props.shape[0], _lib.ptr(u_d))
    rc = lib.fs_shell_quad_fint_noalloc(*args, 16384 + 1, _lib.ptr(fe), None, _lib.stream_ptr())
    assert rc == -22 and b"FS_FINT_NOALLOC_MAX_E" in lib.fs_last_error()
    assert lib.fs_shell_quad_fint_noalloc(*args, 0, _lib.ptr(fe), None, _lib.stream_ptr()) == 0
    assert lib.fs_shell_quad_fint_noalloc(args[0], args[1], args[2], args[3], None, 16, _lib.ptr(fe), None,
                                          _lib.stream_ptr()) == -22
    # the full-size case runs (E = limit) and agrees with the existing entry point
    E = 16384
    props, xyz, pid, u = _inputs(E, 2)
    xyz_d, pid_d, u_d = (torch.from_numpy(a).to(dev) for a in (xyz, pid, u))
    ref, _ = _existing(lib, xyz_d, pid_d, props_d, props.shape[0], u_d, E)
    got, _ = _noalloc(lib, xyz_d, pid_d, props_d, props.shape[0], u_d, E)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_noalloc_fint_under_graph_capture():
    """The use case the entry point exists for: after the set-up call (fs_device_info on the current device resolves
    the workspace address) it is captured in a HIP graph -- nothing in it allocates or resolves symbols -- and the
    replays reproduce the eager result bit for bit, also with new inputs written into the captured buffers."""
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    name = ctypes.create_string_buffer(64)
    cus = ctypes.c_int()
    assert lib.fs_device_info(torch.cuda.current_device(), name, 64, ctypes.byref(cus)) == 0
    E = 2000
    props, xyz, pid, u = _inputs(E, 9)
    props_d = to_device_props(props)
    xyz_d, pid_d, u_d = (torch.from_numpy(a).to(dev) for a in (xyz, pid, u))
    fe = torch.zeros((E, 24), dtype=torch.float64, device=dev)
    st = torch.zeros(E, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        _lib.check(lib.fs_shell_quad_fint_noalloc(_lib.ptr(xyz_d), _lib.ptr(pid_d), _lib.ptr(props_d), props.shape[0],
                                                  _lib.ptr(u_d), E, _lib.ptr(fe), _lib.ptr(st), _lib.stream_ptr()),
                   "fs_shell_quad_fint_noalloc (capture)")
    for seed in (9, 10):
        _, _, _, u2 = _inputs(E, seed)
        u_d.copy_(torch.from_numpy(u2).to(dev))
        fe.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ref, _ = _existing(lib, xyz_d, pid_d, props_d, props.shape[0], u_d, E)
        torch.cuda.synchronize()
        assert torch.equal(fe, ref)
