"""Symbolic phase of the CSR assembly (pattern + scatter plan), computed once per mesh.

Restates MeshAssembler._precompute_elements' DOF map and _compute_sparsity_pattern
(core/assembler.py:166-187, 222-241 of the reference) as array operations on the node graph
instead of per-DOF Python sets: for meshes whose nodes all carry d DOFs the pattern of row d*I+k
is {d*J+l : J neighbour of I, l<d}, so it is fully described by the node-pair blocks (I,J).

Outputs (host NumPy; moved to the device by the assembler):
  nbr_ptr[N+1], nbr[nblk]   node graph in CSR form, neighbours sorted (includes I itself)
  blk_row[nblk]             block row of each block
  blk_ptr[nblk+1], contrib  per block, the (source block index << 1 | transposed) of every
                            contributing element block in FS_LAYOUT_SYMBLK, element order
"""
import numpy as np


BLK_DESC_DTYPE = np.dtype([("dst", "i8"), ("stride", "i4"), ("cnt", "i4"), ("c", "i8", (4,))])
assert BLK_DESC_DTYPE.itemsize == 48
FS_BLK_PACKED = 0x40000000


def pair_index(nn):
    """p(i,j) for i<=j in FS_LAYOUT_SYMBLK and the [nn,nn] lookup (transposed flag for i>j)."""
    P = np.zeros((nn, nn), dtype=np.int64)
    T = np.zeros((nn, nn), dtype=np.int64)
    for i in range(nn):
        for j in range(nn):
            a, b = (i, j) if i <= j else (j, i)
            P[i, j] = a * nn - (a * (a - 1)) // 2 + (b - a)
            T[i, j] = int(i > j)
    return P, T


_T16_RG = np.array([0, 4, 8, 7, 1, 5, 9, 2, 6, 3])     # pair p -> round * 4 + task (include/femshell_hip.h)


def blk_t16(e, p):
    """Block index of (element e, pair p) in FS_LAYOUT_SYMBLK_T16 (include/femshell_hip.h)."""
    e, p = np.asarray(e, dtype=np.int64), np.asarray(p, dtype=np.int64)
    rg = _T16_RG[p]
    r, g, Q = rg >> 2, rg & 3, e & 15
    pos = np.where(r < 2, r * 64 + ((Q >> 3) << 5) + (g << 3) + (Q & 7), 128 + (g << 4) + Q)
    return (e >> 4) * 160 + pos


def can_tile(conn, d):
    """FS_LAYOUT_SYMBLK_T16 applies to uniform quad shell meshes."""
    conn = np.asarray(conn)
    return d == 6 and conn.ndim == 2 and conn.shape[1] == 4 and (conn.shape[0] == 0 or conn.min() >= 0)


def split_groups(conn):
    """Padded connectivity [E, nnmax] (-1 = absent, trailing) -> list of (nn, pos, conn_g):
    pos = positions of the group's elements in the mesh's element order (None: all, in order)."""
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    E, nnmax = conn.shape
    if E == 0 or conn.min() >= 0:
        return [(nnmax, None, conn)]
    cnt = (conn >= 0).sum(axis=1)
    col = np.arange(nnmax)
    if not np.array_equal(conn >= 0, col[None, :] < cnt[:, None]):
        raise ValueError("padded connectivity: -1 entries must trail the node list")
    out = []
    for nn in np.unique(cnt):
        pos = np.flatnonzero(cnt == nn).astype(np.int64)
        out.append((int(nn), pos, np.ascontiguousarray(conn[pos, :nn])))
    if len(out) == 1 and out[0][2].shape[0] == E:
        return [(out[0][0], None, out[0][2])]
    return out


class BlockPlan:
    """Node-pair block scatter plan (uniform d dofs/node; one or more element groups, each with a
    uniform node count; the groups' FS_LAYOUT_SYMBLK buffers are laid out back to back)."""

    def __init__(self, conn: np.ndarray, n_nodes: int, d: int, extra_pairs=None, tiled=False):
        """conn: [E,nn] (or -1-padded for mixed node counts).  extra_pairs: optional int64 [n,2]
        node pairs (I,J) that must exist in the pattern even without a local contribution (ghost
        columns of interface rows in a distributed run).  tiled: the element matrices are stored in
        FS_LAYOUT_SYMBLK_T16 (uniform quad meshes only)."""
        groups = split_groups(conn)
        if tiled and not can_tile(conn, d):
            raise ValueError("FS_LAYOUT_SYMBLK_T16 needs a uniform quad mesh with 6 dofs per node")
        self.tiled = bool(tiled)
        self.n_nodes, self.d = int(n_nodes), int(d)
        self.E = int(np.asarray(conn).shape[0])
        self.groups = []          # (nn, pos, E_g, block offset)
        keys_l, src_l, ord_l = [], [], []
        off = 0
        for nn, pos, cg in groups:
            Eg = cg.shape[0]
            npair = nn * (nn + 1) // 2
            if Eg and (cg.min() < 0 or cg.max() >= n_nodes):
                raise ValueError("connectivity refers to nodes outside [0, n_nodes)")
            # (I, J) key of every local node pair, e-major so a stable sort keeps element order
            keys_l.append((cg[:, :, None] * n_nodes + cg[:, None, :]).reshape(-1))
            self.groups.append((nn, pos, Eg, off))
            off += Eg * npair
        self.n_src_blocks = off if not self.tiled else -(-self.E // 16) * 160
        self.nn = groups[0][0] if len(groups) == 1 else None
        if len(groups) == 1:
            keys = keys_l[0]
            order = np.argsort(keys, kind="stable")
        else:
            keys = np.concatenate(keys_l)
            eord = np.concatenate([np.repeat(pos if pos is not None else np.arange(Eg), nn * nn)
                                   for (nn, pos, Eg, _) in self.groups])
            order = np.lexsort((eord, keys))
            del eord
        skeys = keys[order]
        first = np.ones(skeys.shape[0], dtype=bool)
        first[1:] = skeys[1:] != skeys[:-1]
        ukeys = skeys[first]
        if extra_pairs is not None and len(extra_pairs):
            ep = np.asarray(extra_pairs, dtype=np.int64)
            ukeys = np.union1d(ukeys, ep[:, 0] * n_nodes + ep[:, 1])
        nblk = ukeys.shape[0]
        self.nblk = nblk
        self.blk_row = (ukeys // n_nodes).astype(np.int32)
        self.nbr = (ukeys % n_nodes).astype(np.int32)
        self.nbr_ptr = np.searchsorted(self.blk_row, np.arange(n_nodes + 1), side="left").astype(np.int64)
        self.blk_ptr = np.append(np.searchsorted(skeys, ukeys, side="left"), skeys.shape[0]).astype(np.int64)
        # source block (index << 1 | transposed) of every sorted entry
        contrib = np.empty(order.shape[0], dtype=np.int64)
        lo = 0
        for nn, pos, Eg, boff in self.groups:
            hi = lo + Eg * nn * nn
            sel = (order >= lo) & (order < hi) if len(self.groups) > 1 else slice(None)
            o = order[sel] - lo
            P, T = pair_index(nn)
            ab = o % (nn * nn)
            if self.tiled:
                src_blk = blk_t16(o // 16, P.reshape(-1)[ab])
            else:
                src_blk = boff + (o // (nn * nn)) * (nn * (nn + 1) // 2) + P.reshape(-1)[ab]
            contrib[sel] = (src_blk << 1) | T.reshape(-1)[ab]
            lo = hi
        self.contrib = contrib

    @property
    def nnz(self):
        return self.nblk * self.d * self.d

    def to_device(self, device):
        """Device copies of the plan arrays (+ fs_blk_desc rows as a byte tensor)."""
        import torch
        out = {k: torch.from_numpy(getattr(self, k)).to(device)
               for k in ("nbr_ptr", "nbr", "blk_row", "blk_ptr", "contrib")}
        out["desc"] = torch.from_numpy(self.descriptors().view(np.uint8).reshape(-1)).to(device)
        return out

    def descriptors(self):
        """fs_blk_desc rows (include/femshell_hip.h): output position + inline contributions."""
        d = self.d
        m = np.diff(self.nbr_ptr)
        r0 = self.nbr_ptr[self.blk_row]
        desc = np.zeros(self.nblk, dtype=BLK_DESC_DTYPE)
        desc["dst"] = d * d * r0 + d * (np.arange(self.nblk) - r0)
        desc["stride"] = d * m[self.blk_row]
        cnt = np.diff(self.blk_ptr)
        desc["cnt"] = cnt
        start = self.blk_ptr[:-1]
        for c in range(4):
            has = (cnt > c) & (cnt <= 4)
            desc["c"][has, c] = self.contrib[start[has] + c]
        big = cnt > 4
        desc["c"][big, 0] = start[big]
        # FS_BLK_PACKED: 5..8 contributions whose codes all fit 32 bits sit inline as uint32[8]
        mid = np.flatnonzero(big & (cnt <= 8))
        if mid.size:
            codes = np.zeros((mid.size, 8), dtype=np.int64)
            for c in range(8):
                has = cnt[mid] > c
                codes[has, c] = self.contrib[start[mid][has] + c]
            fits = ((codes >= 0) & (codes < (1 << 32))).all(axis=1)
            m2 = mid[fits]
            desc["c"][m2] = codes[fits].astype(np.uint32).view(np.int64).reshape(-1, 4)
            desc["cnt"][m2] = cnt[m2] | FS_BLK_PACKED
        return desc

    def row_nnz(self):
        """`_row_nnz` of the reference (core/assembler.py:241): int32 [n_nodes*d]."""
        m = np.diff(self.nbr_ptr)
        return np.repeat(m * self.d, self.d).astype(np.int32)

    def host_pattern(self):
        """indptr (int64) / indices (int32) on the host (tests, small meshes)."""
        d = self.d
        m = np.diff(self.nbr_ptr)
        indptr = np.zeros(self.n_nodes * d + 1, dtype=np.int64)
        indptr[1:] = np.cumsum(np.repeat(m * d, d))
        indices = np.empty(self.nnz, dtype=np.int32)
        r0 = self.nbr_ptr[self.blk_row]
        mb = m[self.blk_row]
        base = d * d * r0 + d * (np.arange(self.nblk) - r0)
        for k in range(d):
            for l in range(d):
                indices[base + k * d * mb + l] = self.nbr * d + l
        return indptr, indices


class DeviceBlockPlan:
    """The same plan as BlockPlan, built on the GPU by fs_csr_symbolic_begin / _finish (stable radix
    sort of node-pair keys generated in element order).  The arrays are device tensors; `host()`
    gives a BlockPlan view for the host-side consumers (symmetric descriptors, tests)."""

    def __init__(self, conn: np.ndarray, n_nodes: int, d: int, device, extra_pairs=None, tiled=False):
        import ctypes
        import torch
        from . import _lib
        lib = _lib.load()
        groups = split_groups(conn)
        conn = np.ascontiguousarray(conn, dtype=np.int64)
        if tiled and not can_tile(conn, d):
            raise ValueError("FS_LAYOUT_SYMBLK_T16 needs a uniform quad mesh with 6 dofs per node")
        self.tiled = bool(tiled)
        self.n_nodes, self.d, self.E = int(n_nodes), int(d), int(conn.shape[0])
        self.device = torch.device(device)
        self.groups, off = [], 0
        for nn, pos, cg in groups:
            self.groups.append((nn, pos, cg.shape[0], off))
            off += cg.shape[0] * (nn * (nn + 1) // 2)
        self.n_src_blocks = off if not self.tiled else -(-self.E // 16) * 160
        self.nn = groups[0][0] if len(groups) == 1 else None
        nnmax = conn.shape[1]
        entry_ptr = src_base = None
        n_entries = self.E * nnmax * nnmax
        if len(groups) > 1:                      # ragged rows: per-element entry offsets and source bases
            cnt = (conn >= 0).sum(axis=1)
            ep = np.zeros(self.E + 1, dtype=np.int64)
            np.cumsum(cnt * cnt, out=ep[1:])
            sb = np.empty(self.E, dtype=np.int64)
            for nn, pos, Eg, boff in self.groups:
                sb[pos] = boff + np.arange(Eg, dtype=np.int64) * (nn * (nn + 1) // 2)
            n_entries = int(ep[-1])
            entry_ptr, src_base = torch.from_numpy(ep).to(self.device), torch.from_numpy(sb).to(self.device)
        conn_d = torch.from_numpy(conn).to(self.device)
        extra_d, n_extra = None, 0
        if extra_pairs is not None and len(extra_pairs):
            ep2 = np.asarray(extra_pairs, dtype=np.int64)
            extra_d = torch.from_numpy(np.ascontiguousarray(ep2[:, 0] * self.n_nodes + ep2[:, 1])).to(self.device)
            n_extra = int(extra_d.numel())
        stream = _lib.stream_ptr()
        handle, nblk, ncontrib = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.fs_csr_symbolic_begin(_lib.ptr(conn_d), _lib.ptr(entry_ptr), _lib.ptr(src_base), self.E,
                                             nnmax, _lib.FS_LAYOUT_SYMBLK_T16 if self.tiled else _lib.FS_LAYOUT_SYMBLK,
                                             n_entries, self.n_nodes, _lib.ptr(extra_d), n_extra,
                                             ctypes.byref(handle), ctypes.byref(nblk), ctypes.byref(ncontrib),
                                             stream), "fs_csr_symbolic_begin")
        self.nblk = int(nblk.value)
        i64, i32 = torch.int64, torch.int32
        self.nbr_ptr = torch.empty(self.n_nodes + 1, dtype=i64, device=self.device)
        self.nbr = torch.empty(self.nblk, dtype=i32, device=self.device)
        self.blk_row = torch.empty(self.nblk, dtype=i32, device=self.device)
        self.blk_ptr = torch.empty(self.nblk + 1, dtype=i64, device=self.device)
        self.contrib = torch.empty(int(ncontrib.value), dtype=i64, device=self.device)
        self.desc = torch.empty(self.nblk * BLK_DESC_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        _lib.check(lib.fs_csr_symbolic_finish(handle, self.d, _lib.ptr(self.nbr_ptr), _lib.ptr(self.nbr),
                                              _lib.ptr(self.blk_row), _lib.ptr(self.blk_ptr), _lib.ptr(self.contrib),
                                              _lib.ptr(self.desc), stream), "fs_csr_symbolic_finish")
        self._host = None

    @property
    def nnz(self):
        return self.nblk * self.d * self.d

    def to_device(self, device):
        return {k: getattr(self, k) for k in ("nbr_ptr", "nbr", "blk_row", "blk_ptr", "contrib", "desc")}

    def row_nnz(self):
        m = (self.nbr_ptr[1:] - self.nbr_ptr[:-1]).cpu().numpy()
        return np.repeat(m * self.d, self.d).astype(np.int32)

    def host(self) -> "BlockPlan":
        if self._host is None:
            h = BlockPlan.__new__(BlockPlan)
            h.n_nodes, h.d, h.E, h.groups, h.tiled = self.n_nodes, self.d, self.E, self.groups, self.tiled
            h.n_src_blocks, h.nn, h.nblk = self.n_src_blocks, self.nn, self.nblk
            for k in ("nbr_ptr", "nbr", "blk_row", "blk_ptr", "contrib"):
                setattr(h, k, getattr(self, k).cpu().numpy())
            self._host = h
        return self._host

    def descriptors(self):
        return np.frombuffer(self.desc.cpu().numpy().tobytes(), dtype=BLK_DESC_DTYPE)

    def host_pattern(self):
        return self.host().host_pattern()


def generic_plan(dofs_list, sizes, ndof):
    """Generic per-entry plan for mixed meshes: returns indptr, indices, seg_ptr, src.

    dofs_list: list of (dofs int64 [E_g, nd_g], global element order int64 [E_g]) per element group;
    sizes: offset of each group's dense [E_g, nd_g, nd_g] matrices in the concatenated value buffer.
    Summation order inside a slot == order of appearance == the reference's insertion order when
    groups are concatenated in element order (core/assembler.py:287-300).
    """
    keys_all, src_all, ord_all = [], [], []
    for dofs, base, order in dofs_list_iter(dofs_list, sizes):
        E, nd = dofs.shape
        r = np.repeat(dofs, nd, axis=1).reshape(-1)
        c = np.tile(dofs, (1, nd)).reshape(-1)
        keys_all.append(r * ndof + c)
        src_all.append(base + np.arange(E * nd * nd, dtype=np.int64))
        ord_all.append(np.repeat(order, nd * nd))
    keys = np.concatenate(keys_all)
    src = np.concatenate(src_all)
    eorder = np.concatenate(ord_all)
    o = np.lexsort((eorder, keys))
    keys, src = keys[o], src[o]
    first = np.ones(keys.shape[0], dtype=bool)
    first[1:] = keys[1:] != keys[:-1]
    ukeys = keys[first]
    seg_ptr = np.append(np.flatnonzero(first), keys.shape[0]).astype(np.int64)
    urow = ukeys // ndof
    indices = (ukeys % ndof).astype(np.int32)
    indptr = np.searchsorted(urow, np.arange(ndof + 1), side="left").astype(np.int64)
    return indptr, indices, seg_ptr, src


def dofs_list_iter(dofs_list, sizes):
    for (dofs, order), base in zip(dofs_list, sizes):
        yield np.ascontiguousarray(dofs, dtype=np.int64), int(base), np.asarray(order, dtype=np.int64)


def vector_plan(conn, n_nodes, d):
    """Segmented gather plan for vectors (f_int, load vectors): for DOF r the entries
    src[seg_ptr[r]:seg_ptr[r+1]] index the element vectors ([E_g, nn*d] per group, the groups'
    buffers back to back), in element order (== the reference's setValuesLocal insertion order,
    core/assembler.py:369-381)."""
    groups = split_groups(conn)
    dofs_l, src_l, ord_l = [], [], []
    off = 0
    for nn, pos, cg in groups:
        Eg = cg.shape[0]
        dofs_l.append((cg[:, :, None] * d + np.arange(d, dtype=np.int64)).reshape(-1))
        src_l.append(off + np.arange(Eg * nn * d, dtype=np.int64))
        ord_l.append(np.repeat(pos if pos is not None else np.arange(Eg, dtype=np.int64), nn * d))
        off += Eg * nn * d
    if len(groups) == 1:
        dofs = dofs_l[0]
        order = np.argsort(dofs, kind="stable").astype(np.int64)
        src = order
    else:
        dofs = np.concatenate(dofs_l)
        order = np.lexsort((np.concatenate(ord_l), dofs))
        src = np.concatenate(src_l)[order]
    seg_ptr = np.searchsorted(dofs[order], np.arange(n_nodes * d + 1), side="left").astype(np.int64)
    return seg_ptr, src
