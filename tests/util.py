"""Shared test helpers: seeded element generators and property tables."""
import numpy as np

import fem_shell_amd as fsa

BASE_QUAD = np.array([[0, 0, 0], [1, 0, .05], [1.1, 1, 0], [0, 1, .1]], float)

T300 = dict(E=(181e9, 10.3e9, 10.3e9), G=(7.17e9, 3.0e9, 7.17e9), nu=(0.28, 0.28, 0.28), rho=1600.0)


def t300():
    return fsa.OrthotropicMaterial("T300/5208", T300["E"], T300["G"], T300["nu"], T300["rho"])


def standard_props():
    """A mixed property table: isotropic (Ko-2017 materials), symmetric and asymmetric laminates."""
    m = t300()
    lam = lambda ang, t=0.125e-3: fsa.create_laminate_from_angles(m, t, ang)
    return [
        fsa.ShellProperty(fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0), 0.01),
        fsa.ShellProperty(fsa.IsotropicMaterial("scordelis", 4.32e8, 0.0, 360.0), 0.25),
        fsa.ShellProperty(fsa.IsotropicMaterial("hemi", 6.825e7, 0.3, 1.0), 0.04),
        fsa.CompositeShellProperty(lam([0, 90, 90, 0])),
        fsa.CompositeShellProperty(lam([0, 90])),
        fsa.CompositeShellProperty(lam([0, 45, 90])),
        fsa.CompositeShellProperty(lam([30], 1e-3)),
        fsa.CompositeShellProperty(lam([0, 45, -45, 90, 90, -45, 45, 0], 2.5e-3)),
    ]


def random_quads(n, seed=0, far=True):
    """Warped / sheared / scaled / translated quads (the L/h-stressing cases of SURVEY.md A.1)."""
    rng = np.random.default_rng(seed)
    xyz = np.empty((n, 4, 3))
    for t in range(n):
        off = rng.choice([0, 0, 1e3, 3.0] if far else [0.0]) * rng.standard_normal(3)
        scale = rng.choice([1, 1e-3, 20])
        q = BASE_QUAD + 0.1 * rng.standard_normal((4, 3))
        if t % 7 == 0:
            q = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], float)       # unit square
        if t % 7 == 1:
            q = np.array([[0, 0, 0], [2, 0, 0], [2.5, 1, 0], [.5, 1, 0]], float)    # parallelogram
        xyz[t] = q * scale + off
    return xyz


def pack_symblk(dense, nn=4, d=6, tiled=False):
    """dense [E, nn*d, nn*d] -> FS_LAYOUT_SYMBLK [E, npair, d*d], or FS_LAYOUT_SYMBLK_T16
    [ceil(E/16)*160, 36] with tiled=True (host, tests only)."""
    from fem_shell_amd.symbolic import blk_t16, pair_index
    P, _ = pair_index(nn)
    E = dense.shape[0]
    out = np.empty((E, nn * (nn + 1) // 2, d * d))
    for i in range(nn):
        for j in range(i, nn):
            out[:, P[i, j]] = dense[:, d * i:d * i + d, d * j:d * j + d].reshape(-1, d * d)
    if tiled:
        t16 = np.zeros((-(-E // 16) * 160, d * d))
        for p in range(10):
            t16[blk_t16(np.arange(E), p)] = out[:, p]
        return t16
    return out


def host_block_assemble(plan, packed):
    """NumPy emulation of fs_csr_numeric_desc (same element-order summation); tests only."""
    d = plan.d
    flat = packed.reshape(-1, d, d)
    cnt = np.diff(plan.blk_ptr)
    acc = np.zeros((plan.nblk, d, d))
    for c in range(int(cnt.max()) if cnt.size else 0):
        has = cnt > c
        cc = plan.contrib[plan.blk_ptr[:-1][has] + c]
        blk = flat[cc >> 1]
        blk = np.where((cc & 1)[:, None, None].astype(bool), blk.transpose(0, 2, 1), blk)
        acc[has] = acc[has] + blk
    m = np.diff(plan.nbr_ptr)
    r0 = plan.nbr_ptr[plan.blk_row]
    base = d * d * r0 + d * (np.arange(plan.nblk) - r0)
    stride = d * m[plan.blk_row]
    idx = base[:, None, None] + np.arange(d)[None, :, None] * stride[:, None, None] + np.arange(d)[None, None, :]
    data = np.zeros(plan.nnz)
    data[idx.reshape(-1)] = acc.reshape(-1)
    return data


FUSED_RTOL = 2e-14      # fused quad kernel against element kernel + assembly kernel, relative to the largest entry


def assert_fused_close(Kf, K2, what="K", rtol=FUSED_RTOL):
    """The fused MITC4+ stiffness route against the two-kernel route.  Since round 3 the fused kernel forms a contribution
    that its block receives transposed as the block of the swapped node pair (instead of transposing it in registers)
    and keeps no signed-zero bookkeeping, so the two routes agree to rounding (measured: 1.2e-16 of the largest entry),
    not to the bit; each route by itself is bitwise repeatable, and the contract is 1e-12 against the reference."""
    import torch
    scale = float(K2.data.abs().max())
    d = float((Kf.data - K2.data).abs().max())
    assert not bool(torch.isnan(Kf.data).any()) and d <= rtol * scale, (what, d / scale)
    return d / scale


def assert_exactly_symmetric(K):
    """K == K^T entry by entry (the fused kernels write block (J, I) from the registers of block (I, J))."""
    A = K.to_scipy()
    assert (A != A.T).nnz == 0
