"""Array-native batch front-end of the C ABI: torch tensors in, torch tensors out (device
buffers only -- all arithmetic is in libfemshell_hip.so).  This is what MeshAssembler and the
per-element classes delegate to; no per-element Python objects are created here."""
import numpy as np
import torch

from . import _lib
from .properties import PROP_DTYPE

NPAIR4 = 10


def _dev():
    if not torch.cuda.is_available():
        raise _lib.FemShellHipError("no GPU visible: the fem_shell_amd hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_props(props: np.ndarray) -> torch.Tensor:
    props = np.ascontiguousarray(props)
    assert props.dtype == PROP_DTYPE
    raw = torch.from_numpy(props.view(np.uint8).reshape(-1).copy())
    return raw.to(_dev())


def n_blocks(E, npair, layout=_lib.FS_LAYOUT_SYMBLK):
    """Blocks in the element-matrix buffer of E elements (FS_LAYOUT_SYMBLK_T16 pads to whole tiles)."""
    return -(-E // 16) * 160 if layout == _lib.FS_LAYOUT_SYMBLK_T16 else E * npair


def _out_buf(given, E, npair, dev, dd=36, layout=_lib.FS_LAYOUT_SYMBLK):
    """Caller-provided output view (element groups of one mesh share a block buffer) or a new one."""
    if layout == _lib.FS_LAYOUT_SYMBLK_T16:
        nb = n_blocks(E, npair, layout)
        if given is None:      # the padding of the last tile is never read: keep it finite anyway
            return torch.zeros((nb, dd), dtype=torch.float64, device=dev)
        assert given.dtype == torch.float64 and given.is_cuda and given.is_contiguous() and given.numel() == nb * dd
        return given.view(nb, dd)
    if given is None:
        return torch.empty((E, npair, dd), dtype=torch.float64, device=dev)
    assert given.dtype == torch.float64 and given.is_cuda and given.is_contiguous()
    assert given.numel() == E * npair * dd
    return given.view(E, npair, dd)


def quad_batch(xyz, prop_id, props_dev, n_props, want_K=True, want_M=False, dense=False,
               status=True, validate=True, u_e=None, want_KT=False, want_F=False, K_out=None,
               M_out=None, KT_out=None, layout=_lib.FS_LAYOUT_SYMBLK, K_lin=None):
    """MITC4+ element matrices for E quads.

    K_lin: linear blocks K of the same elements from an earlier want_K call (tiled layout): KT / f_int then skip the
    stiffness kernel (they do not depend on u_e; the reference reads its cached MITC4.K).

    xyz [E,4,3] f64 device, prop_id [E] i32 device, props_dev = to_device_props(table).
    Returns dict with 'K'/'M' (FS_LAYOUT_SYMBLK [E,10,36] -- or FS_LAYOUT_SYMBLK_T16 [ceil(E/16)*160,36]
    with layout=_lib.FS_LAYOUT_SYMBLK_T16 -- or dense [E,24,24]) and 'status'.
    """
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    assert xyz.dtype == torch.float64 and xyz.shape[1:] == (4, 3) and xyz.is_cuda
    assert prop_id.dtype == torch.int32 and prop_id.is_cuda and prop_id.numel() == E
    if E and validate:   # host-synchronising range check; callers in hot loops validate once
        pmin, pmax = int(prop_id.min()), int(prop_id.max())
        if pmin < 0 or pmax >= n_props:
            raise ValueError("prop_id out of range of the property table")
    out = {}
    K = _out_buf(K_out, E, NPAIR4, dev, layout=layout) if want_K else None
    M = _out_buf(M_out, E, NPAIR4, dev, layout=layout) if want_M else None
    st = torch.zeros(E, dtype=torch.int32, device=dev) if status else None
    want = (_lib.FS_WANT_K if want_K else 0) | (_lib.FS_WANT_M if want_M else 0)
    KT = F = None
    if want_KT or want_F:
        if u_e is None:
            raise ValueError("KT / f_int need the element displacement vectors u_e [E,24]")
        u_e = u_e.contiguous()
        assert u_e.dtype == torch.float64 and u_e.shape == (E, 24) and u_e.is_cuda
        cached = K_lin is not None and not want_K and layout == _lib.FS_LAYOUT_SYMBLK_T16
        if cached:
            assert K_lin.dtype == torch.float64 and K_lin.is_cuda and K_lin.is_contiguous() and K_lin.numel() >= -(-E // 16) * 160 * 36
        if want_KT or not cached:
            KT = _out_buf(KT_out, E, NPAIR4, dev, layout=layout)   # also the FINT workspace without K_lin
        F = torch.empty((E, 24), dtype=torch.float64, device=dev) if want_F else None
        want |= (_lib.FS_WANT_KT if want_KT else 0) | (_lib.FS_WANT_FINT if want_F else 0)
        if cached:
            K = K_lin
    _lib.check(lib.fs_shell_quad_batch(_lib.ptr(xyz), _lib.ptr(prop_id), _lib.ptr(props_dev),
                                       n_props, _lib.ptr(u_e), E, want, layout,
                                       _lib.ptr(K), _lib.ptr(M), _lib.ptr(KT), _lib.ptr(F), _lib.ptr(st),
                                       _lib.stream_ptr()), "fs_shell_quad_batch")
    if not want_K:
        K = None
    if not want_KT:
        KT = None
    if dense:
        K = expand_blocks(K, 4, 6, layout, E) if K is not None else None
        M = expand_blocks(M, 4, 6, layout, E) if M is not None else None
        KT = expand_blocks(KT, 4, 6, layout, E) if KT is not None else None
    out["K"], out["M"], out["KT"], out["f_int"], out["status"] = K, M, KT, F, st
    return out


NPAIR3 = 6


def tri_batch(xyz, prop_id, props_dev, n_props, want_K=True, want_M=False, dense=False,
              status=True, validate=True, K_out=None, M_out=None, u_e=None, want_KT=False,
              want_F=False, KT_out=None, K_lin=None):
    """MITC3+ element matrices for E triangles (fs_shell_tri_batch).  K_lin: cached linear blocks, see quad_batch.

    xyz [E,3,3] f64 device.  Returns dict with 'K'/'M'/'KT' (FS_LAYOUT_SYMBLK [E,6,36] or dense
    [E,18,18]), 'f_int' [E,18] and 'status'."""
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    assert xyz.dtype == torch.float64 and xyz.shape[1:] == (3, 3) and xyz.is_cuda
    assert prop_id.dtype == torch.int32 and prop_id.is_cuda and prop_id.numel() == E
    if E and validate:
        pmin, pmax = int(prop_id.min()), int(prop_id.max())
        if pmin < 0 or pmax >= n_props:
            raise ValueError("prop_id out of range of the property table")
    K = _out_buf(K_out, E, NPAIR3, dev) if want_K else None
    M = _out_buf(M_out, E, NPAIR3, dev) if want_M else None
    st = torch.zeros(E, dtype=torch.int32, device=dev) if status else None
    want = (_lib.FS_WANT_K if want_K else 0) | (_lib.FS_WANT_M if want_M else 0)
    KT = F = None
    if want_KT or want_F:
        if u_e is None:
            raise ValueError("KT / f_int need the element displacement vectors u_e [E,18]")
        u_e = u_e.contiguous()
        assert u_e.dtype == torch.float64 and u_e.shape == (E, 18) and u_e.is_cuda
        cached = K_lin is not None and not want_K
        if cached:
            assert K_lin.dtype == torch.float64 and K_lin.is_cuda and K_lin.is_contiguous() and K_lin.numel() == E * NPAIR3 * 36
            K = K_lin
        if want_KT or not cached:
            KT = _out_buf(KT_out, E, NPAIR3, dev)   # also the FINT workspace without K_lin
        F = torch.empty((E, 18), dtype=torch.float64, device=dev) if want_F else None
        want |= (_lib.FS_WANT_KT if want_KT else 0) | (_lib.FS_WANT_FINT if want_F else 0)
    _lib.check(lib.fs_shell_tri_batch(_lib.ptr(xyz), _lib.ptr(prop_id), _lib.ptr(props_dev),
                                      n_props, _lib.ptr(u_e), E, want, _lib.FS_LAYOUT_SYMBLK,
                                      _lib.ptr(K), _lib.ptr(M), _lib.ptr(KT), _lib.ptr(F), _lib.ptr(st),
                                      _lib.stream_ptr()), "fs_shell_tri_batch")
    if not want_KT:
        KT = None
    if not want_K:
        K = None
    if dense:
        K = expand_blocks(K, 3, 6) if K is not None else None
        M = expand_blocks(M, 3, 6) if M is not None else None
        KT = expand_blocks(KT, 3, 6) if KT is not None else None
    return {"K": K, "M": M, "KT": KT, "f_int": F, "status": st}


def shell_kg_batch(nn, xyz, prop_id, props_dev, n_props, sigma, dense=False, KG_out=None,
                   layout=_lib.FS_LAYOUT_SYMBLK):
    """Shell geometric stiffness from local membrane stresses sigma [E,3] (fs_shell_kg_batch):
    FS_LAYOUT_SYMBLK [E, nn(nn+1)/2, 36] or dense [E,6nn,6nn]."""
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    if nn not in (3, 4):
        raise ValueError(f"unsupported shell element with {nn} nodes")
    assert xyz.dtype == torch.float64 and xyz.shape[1:] == (nn, 3) and xyz.is_cuda
    assert prop_id.dtype == torch.int32 and prop_id.is_cuda and prop_id.numel() == E
    sigma = sigma.contiguous()
    assert sigma.dtype == torch.float64 and sigma.shape == (E, 3) and sigma.is_cuda
    KG = _out_buf(KG_out, E, nn * (nn + 1) // 2, dev, layout=layout)
    _lib.check(lib.fs_shell_kg_batch(nn, _lib.ptr(xyz), _lib.ptr(prop_id), _lib.ptr(props_dev), n_props,
                                     _lib.ptr(sigma), E, layout, _lib.ptr(KG),
                                     _lib.stream_ptr()), "fs_shell_kg_batch")
    return expand_blocks(KG, nn, 6, layout, E) if dense else KG


def shell_kt_sigma_batch(nn, xyz, prop_id, props_dev, n_props, u_e, sigma):
    """K_T with a given local membrane stress sigma [E,3] (fs_shell_kt_sigma_batch) -> dense [E,6nn,6nn], status."""
    lib = _lib.load()
    dev = _dev()
    xyz, u_e, sigma = xyz.contiguous(), u_e.contiguous(), sigma.contiguous()
    E = xyz.shape[0]
    if nn not in (3, 4):
        raise ValueError(f"unsupported shell element with {nn} nodes")
    assert xyz.dtype == torch.float64 and xyz.shape[1:] == (nn, 3) and xyz.is_cuda
    assert prop_id.dtype == torch.int32 and prop_id.is_cuda and prop_id.numel() == E
    assert u_e.dtype == torch.float64 and u_e.shape == (E, 6 * nn) and u_e.is_cuda
    assert sigma.dtype == torch.float64 and sigma.shape == (E, 3) and sigma.is_cuda
    KT = _out_buf(None, E, nn * (nn + 1) // 2, dev)
    status = torch.zeros(E, dtype=torch.int32, device=dev)
    _lib.check(lib.fs_shell_kt_sigma_batch(nn, _lib.ptr(xyz), _lib.ptr(prop_id), _lib.ptr(props_dev), n_props,
                                           _lib.ptr(u_e), _lib.ptr(sigma), E, _lib.ptr(KT), _lib.ptr(status),
                                           _lib.stream_ptr()), "fs_shell_kt_sigma_batch")
    return expand_blocks(KT, nn, 6, _lib.FS_LAYOUT_SYMBLK, E), status


def shell_rotate_local(nn, xyz, K=None, f=None):
    """Local-axes view (T K T^T, T f) of dense global-axes element results (fs_shell_rotate_local); the frame is the
    kernels' geometry stage on xyz [E,nn,3]."""
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    assert xyz.dtype == torch.float64 and xyz.shape[1:] == (nn, 3) and xyz.is_cuda
    Kl = fl = None
    if K is not None:
        K = K.contiguous()
        assert K.dtype == torch.float64 and K.shape == (E, 6 * nn, 6 * nn) and K.is_cuda
        Kl = torch.empty_like(K)
    if f is not None:
        f = f.contiguous()
        assert f.dtype == torch.float64 and f.shape == (E, 6 * nn) and f.is_cuda
        fl = torch.empty_like(f)
    _lib.check(lib.fs_shell_rotate_local(nn, _lib.ptr(xyz), E, _lib.ptr(K), _lib.ptr(f), _lib.ptr(Kl), _lib.ptr(fl),
                                         _lib.stream_ptr()), "fs_shell_rotate_local")
    return Kl, fl


def centrifugal_prestress(solid, nn, xyz, prop_id, props_dev, n_props, omega, rotation_axis, rotation_center):
    """compute_centrifugal_prestress of every element (fs_centrifugal_prestress): [E,3] local
    membrane stresses for MITC3, [E,6] global Voigt stresses for the linear solids."""
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    if not solid and nn == 4:
        raise AttributeError("'MITC4' object has no attribute 'compute_centrifugal_prestress'")
    ax = np.ascontiguousarray(np.asarray(rotation_axis, dtype=np.float64).reshape(3))
    ce = np.ascontiguousarray(np.asarray(rotation_center, dtype=np.float64).reshape(3))
    sig = torch.empty((E, 6 if solid else 3), dtype=torch.float64, device=dev)
    _lib.check(lib.fs_centrifugal_prestress(int(bool(solid)), nn, _lib.ptr(xyz), _lib.ptr(prop_id),
                                            _lib.ptr(props_dev), n_props, float(omega), ax.ctypes.data,
                                            ce.ctypes.data, E, _lib.ptr(sig), _lib.stream_ptr()),
               "fs_centrifugal_prestress")
    return sig


def solid_batch(nn, xyz, prop_id, props_dev, n_props, want_K=True, want_M=False, body=None,
                sigma=None, reduced=False, dense=False, status=True, validate=True, K_out=None,
                M_out=None):
    """Solids (fs_solid_batch): nn = 4 TETRA4, 5 PYRAMID5, 6 WEDGE6, 8 HEXA8, 10 TETRA10, 13 PYRAMID13, 15 WEDGE15, 20 HEXA20.

    xyz [E,nn,3] f64 device; props_dev [P,37] f64 device (properties.solid_row).  body: 3 floats
    (host) -> 'f' [E,3nn]; sigma [E,6] device -> 'KG'.  Matrices in FS_LAYOUT_SYMBLK
    [E, nn(nn+1)/2, 9] or dense [E,3nn,3nn]."""
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    if nn not in (4, 5, 6, 8, 10, 13, 15, 20):
        raise NotImplementedError(f"solid elements with {nn} nodes are not on the GPU path")
    assert xyz.dtype == torch.float64 and xyz.shape[1:] == (nn, 3) and xyz.is_cuda
    assert prop_id.dtype == torch.int32 and prop_id.is_cuda and prop_id.numel() == E
    assert props_dev.dtype == torch.float64 and props_dev.numel() == 37 * n_props
    if E and validate:
        pmin, pmax = int(prop_id.min()), int(prop_id.max())
        if pmin < 0 or pmax >= n_props:
            raise ValueError("prop_id out of range of the property table")
    npair = nn * (nn + 1) // 2
    K = _out_buf(K_out, E, npair, dev, 9) if want_K else None
    M = _out_buf(M_out, E, npair, dev, 9) if want_M else None
    st = torch.zeros(E, dtype=torch.int32, device=dev) if status else None
    want = (_lib.FS_WANT_K if want_K else 0) | (_lib.FS_WANT_M if want_M else 0)
    F = KG = b3 = None
    if body is not None:
        b3 = np.ascontiguousarray(np.asarray(body, dtype=np.float64)[:3])
        F = torch.empty((E, 3 * nn), dtype=torch.float64, device=dev)
        want |= _lib.FS_WANT_FBODY
    if sigma is not None:
        sigma = sigma.contiguous()
        assert sigma.dtype == torch.float64 and sigma.shape == (E, 6) and sigma.is_cuda
        KG = torch.empty((E, npair, 9), dtype=torch.float64, device=dev)
        want |= _lib.FS_WANT_KG
    _lib.check(lib.fs_solid_batch(nn, int(bool(reduced)), _lib.ptr(xyz), _lib.ptr(prop_id),
                                  _lib.ptr(props_dev), n_props, E, want, _lib.FS_LAYOUT_SYMBLK,
                                  b3.ctypes.data if b3 is not None else None, _lib.ptr(sigma),
                                  _lib.ptr(K), _lib.ptr(M), _lib.ptr(F), _lib.ptr(KG), _lib.ptr(st),
                                  _lib.stream_ptr()), "fs_solid_batch")
    if dense:
        K = expand_blocks(K, nn, 3) if K is not None else None
        M = expand_blocks(M, nn, 3) if M is not None else None
        KG = expand_blocks(KG, nn, 3) if KG is not None else None
    return {"K": K, "M": M, "f": F, "KG": KG, "status": st}


def to_device_solid_props(rows) -> torch.Tensor:
    rows = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, 37))
    return torch.from_numpy(rows).to(_dev())


def shell_batch(nn, *args, **kw):
    """Dispatch on nodes per element (4 -> MITC4+, 3 -> MITC3+)."""
    if nn == 4:
        return quad_batch(*args, **kw)
    if nn == 3:
        if kw.pop("layout", _lib.FS_LAYOUT_SYMBLK) != _lib.FS_LAYOUT_SYMBLK:
            raise ValueError("FS_LAYOUT_SYMBLK_T16 is a quad layout")
        return tri_batch(*args, **kw)
    raise ValueError(f"unsupported shell element with {nn} nodes")


def expand_blocks(packed, nn, d, layout=_lib.FS_LAYOUT_SYMBLK, E=None):
    """Element-matrix blocks -> dense [E, nn*d, nn*d] (E must be given for FS_LAYOUT_SYMBLK_T16)."""
    lib = _lib.load()
    if E is None:
        if layout != _lib.FS_LAYOUT_SYMBLK:
            raise ValueError("expand_blocks: pass E for the tiled layout")
        E = packed.shape[0]
    dense = torch.empty((E, nn * d, nn * d), dtype=torch.float64, device=packed.device)
    _lib.check(lib.fs_blk_expand(_lib.ptr(packed), layout, E, nn, d, _lib.ptr(dense), _lib.stream_ptr()),
               "fs_blk_expand")
    return dense


def quad_geom(xyz):
    lib = _lib.load()
    dev = _dev()
    xyz = xyz.contiguous()
    E = xyz.shape[0]
    frame = torch.empty((E, 3, 3), dtype=torch.float64, device=dev)
    xl = torch.empty((E, 4, 2), dtype=torch.float64, device=dev)
    detJ = torch.empty((E, 4), dtype=torch.float64, device=dev)
    acoef = torch.empty((E, 5), dtype=torch.float64, device=dev)
    _lib.check(lib.fs_shell_quad_geom(_lib.ptr(xyz), E, _lib.ptr(frame), _lib.ptr(xl),
                                      _lib.ptr(detJ), _lib.ptr(acoef), _lib.stream_ptr()),
               "fs_shell_quad_geom")
    return frame, xl, detJ, acoef
