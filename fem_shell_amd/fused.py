"""Plan of the fused evaluate-and-assemble kernel (fs_fused_assemble_quad, include/femshell_hip.h).

The kernel replaces the element loop + setValuesLocal loop of the reference (core/assembler.py:160-193, 287-303) for
uniform MITC4+ meshes by ONE launch that never writes element matrices to memory.  It needs, once per mesh:

  * a partition of the nodes into PATCHES such that the elements touching a patch fit the LDS image of one
    workgroup (<= 64, or <= 80, elements): `partition_nodes` bins the nodes on a grid of about 7 x 7 mean element
    edges (a 7 x 7 node patch of a structured mesh touches exactly 64 elements) and bisects patches that are
    still too big -- any mesh works, compact patches just recompute fewer halo elements;
  * per patch, the CSR node-pair blocks it produces -- block (I, J) with I in the patch and I <= J, written to
    (I, J) and transposed to (J, I) -- expanded into one record per lane: every contribution (element, local
    node pair, transposed?) of a block sits on its own lane, the contributions of a block on adjacent lanes of
    a quad in ELEMENT ORDER (the reference's ADD_VALUES insertion order, core/assembler.py:295-300), so the
    kernel's in-quad sum is the sum fs_csr_numeric_desc forms.

Everything here is integer bookkeeping on the symbolic plan (symbolic.BlockPlan); no floating-point work.
"""
import os

import numpy as np

from .symbolic import _T16_RG

REC_DTYPE = np.dtype([("dst", "i8", (2,)), ("dst_t", "i8", (2,)), ("stride", "i4", (2,)), ("stride_t", "i4", (2,)),
                      ("ev", "u4", (2,)), ("flags", "u4"), ("slots", "u4")])
assert REC_DTYPE.itemsize == 64
# ev bits: element slot [0:8) | ti [8:10) | tj [10:12) | transposed 12 | active 13
# flags bits (include/femshell_hip.h FS_FL_*)
FL_CHAIN, FL_OUT0, FL_OUT1, FL_ZERO_START, FL_PAD = 1, 2, 4, 8, 16
FL_MAXCHAIN_SHIFT, FL_POS_SHIFT = 8, 16

# local node pair p -> (ti, tj), ti <= tj, of FS_LAYOUT_SYMBLK (nn = 4)
_PAIR_I = np.array([0, 0, 0, 0, 1, 1, 1, 2, 2, 3], dtype=np.uint32)
_PAIR_J = np.array([0, 1, 2, 3, 1, 2, 3, 2, 3, 3], dtype=np.uint32)


def _t16_inverse():
    """position inside a 160-block tile of FS_LAYOUT_SYMBLK_T16 -> (element within the tile, pair p)."""
    q = np.zeros(160, dtype=np.int64)
    pp = np.zeros(160, dtype=np.int64)
    for p in range(10):
        rg = int(_T16_RG[p])
        r, g = rg >> 2, rg & 3
        for Q in range(16):
            pos = r * 64 + ((Q >> 3) << 5) + (g << 3) + (Q & 7) if r < 2 else 128 + (g << 4) + Q
            q[pos], pp[pos] = Q, p
    return q, pp


_T16_Q, _T16_P = _t16_inverse()


GROUP_NODES = 12     # nodes whose tasks share one wave (5 lanes per interior node of a structured mesh)
GRID_NODES = 6       # nodes per grid cell along the node-numbering direction (a wave = 2 grid rows of 6)


def _numbering_axis(coords, used):
    """Global axis along which consecutive node indices advance (structured meshes: the fastest index)."""
    idx = np.flatnonzero(used)
    if idx.shape[0] < 2:
        return 0
    d = np.abs(coords[idx[1:]] - coords[idx[:-1]])
    return int(np.argmax(np.median(d, axis=0)))


def _lattice(conn, N):
    """(i, j) lattice indices of the nodes if the mesh is numbered like a structured grid (node index = j * L + i: the
    element edges join indices that differ by 1 and by one row length L, whatever the geometry), else None.  Wrap-around
    elements of closed tubes and a few irregular elements are tolerated (the bisection pass repairs their patches)."""
    E = conn.shape[0]
    if E < 4:
        return None
    d01, d32 = np.abs(conn[:, 1] - conn[:, 0]), np.abs(conn[:, 2] - conn[:, 3])
    d03, d12 = np.abs(conn[:, 3] - conn[:, 0]), np.abs(conn[:, 2] - conn[:, 1])
    lo, hi = np.minimum(d01, d03), np.maximum(d01, d03)
    vals, cnts = np.unique(hi, return_counts=True)
    L = int(vals[np.argmax(cnts)])
    if L < 2:
        return None
    ok = (lo == 1) & (hi == L) & (d32 == d01) & (d12 == d03)
    if ok.mean() < 0.9:
        return None
    idx = np.arange(N, dtype=np.int64)
    return np.stack([idx % L, idx // L], axis=1).astype(np.float64)


def _elements_per_patch(patch, conn, npat):
    """Number of elements touching each patch: distinct (patch, element) pairs, counted in O(E) -- the patches of an
    element's nodes are sorted within the element's row and a patch counts where it first appears."""
    pc = np.sort(patch[conn], axis=1)
    first = np.ones(pc.shape, dtype=bool)
    first[:, 1:] = pc[:, 1:] != pc[:, :-1]
    return np.bincount(pc[first], minlength=npat)


# Patch ids are numbered in TILES of PATCH_TILE[0] x PATCH_TILE[1] grid cells (numbering direction x the next one), not row
# by row: the kernel gives the 32 workgroups of an XCD 32 CONSECUTIVE patches at a time, and the CSR lines and halo elements
# a patch shares with its neighbours meet in that XCD's L2 only if the neighbours are at work at about the same time.  Row by
# row, the vertical neighbours of a patch of the 10 M plate come 17 rounds later (534 patches per row) and the shared lines
# travel to HBM twice: 10.13 ms per launch; 4 x 8 tiles 9.64 ms (2 x 8 ... 4 x 8 within 0.3 %, 8 x 2 10.03, whole columns 16 %
# slower on the strip).  The order changes nothing in the result (every CSR entry has one writer).  The MASS operator of the
# same kernel, all stores and hardly any arithmetic, is 6-19 % slower on any tiled order (strip: 0.958 ms row by row, 1.088
# in 4 x 8 tiles), so MeshAssembler keeps a second, row-by-row plan for it (tile=(1, 1)).  FEMSHELL_PATCH_TILE=WxH overrides
# the stiffness order for experiments (tools/exp/tile_full.sh, tile_ab.sh).
PATCH_TILE = tuple(int(v) for v in os.environ.get("FEMSHELL_PATCH_TILE", "4x8").split("x"))


def _tiled_patch_key(bi, bj, nbi, tile=None):
    """Sort key of grid cell (bi, bj) -- NumPy arrays or torch tensors -- in tiles of `tile` (default PATCH_TILE) cells, tiles
    row by row, cells row by row inside a tile; (1, 1) is the plain row-by-row order."""
    tw, th = tile or PATCH_TILE
    ntx = (nbi + tw - 1) // tw
    return ((bj // th) * ntx + (bi // tw)) * (tw * th) + (bj % th) * tw + (bi % tw)


def partition_nodes(coords, conn, max_elems=64, tile=None):
    """node -> patch id (int64 [N]); the elements touching any patch number <= max_elems.

    Grid binning: cells of a x b nodes, a (<= GRID_NODES) along the direction in which consecutive node indices run
    and a b as large as (a + 1)(b + 1) <= max_elems allows (6 x 8 nodes / 63 elements for 64; patch ids follow the
    grid in tiles of PATCH_TILE cells, so that 32 consecutive patches are a compact block of neighbours), then recursive median bisection, along the longest
    extent, of the patches that still touch too many elements.  The grid lives on the (i, j) lattice of the node
    numbering when the mesh is numbered like a structured grid (curved shells: the patches follow the mesh lines),
    otherwise on the coordinates with the mean element edge as spacing.  Any mesh works; compact patches recompute
    fewer halo elements and rows of consecutively numbered nodes give the kernel's stores long contiguous runs."""
    coords = np.asarray(coords, dtype=np.float64)
    conn = np.asarray(conn, dtype=np.int64)
    N, E = coords.shape[0], conn.shape[0]
    if E == 0:
        return np.zeros(N, dtype=np.int64)
    used = np.zeros(N, dtype=bool)
    used[conn.reshape(-1)] = True
    lat = _lattice(conn, N)
    if lat is not None:
        coords, hk, ax = lat, np.ones(2), 0
    else:
        x = coords[conn]
        e1, e2 = np.abs(x[:, 1] - x[:, 0]), np.abs(x[:, 3] - x[:, 0])
        hk = np.maximum(np.maximum(e1, e2).mean(axis=0), 1e-300)          # node spacing seen along each global axis
        hk = np.maximum(hk, 1e-3 * float(hk.max()))
        ax = _numbering_axis(coords, used)
    nd = coords.shape[1]
    best = (1, 1)
    for a in range(1, GRID_NODES + 1):
        for bb in range(1, 40):
            if (a + 1) * (bb + 1) <= max_elems and a * bb > best[0] * best[1]:
                best = (a, bb)
    nodes_per_cell = np.full(nd, float(best[1]))
    nodes_per_cell[ax] = float(best[0])
    side = nodes_per_cell * hk
    lo = coords[used].min(axis=0) - 0.5 * hk
    cc = np.where(used[:, None], coords, lo)          # nodes no element touches (ghost columns) may carry NaN
    b = np.clip(np.floor((cc - lo) / side).astype(np.int64), 0, None)
    nb = b[used].max(axis=0) + 1
    b = np.minimum(b, nb - 1)
    order_axes = [ax] + [k for k in range(nd) if k != ax]             # numbering axis fastest
    if nd == 2:
        key = _tiled_patch_key(b[:, order_axes[0]], b[:, order_axes[1]], int(nb[order_axes[0]]), tile)
    else:
        key = np.zeros(N, dtype=np.int64)
        for k in reversed(order_axes):
            key = key * nb[k] + b[:, k]
    _, patch = np.unique(key, return_inverse=True)
    patch = patch.astype(np.int64)
    for _ in range(64):
        npat = int(patch.max()) + 1
        cnt = _elements_per_patch(patch, conn, npat)
        bad = cnt > max_elems
        if not bad.any():
            return patch
        nodes = np.flatnonzero(bad[patch] & used)
        pb = patch[nodes]
        o = np.argsort(pb, kind="stable")
        nodes, pb = nodes[o], pb[o]
        starts = np.flatnonzero(np.r_[True, pb[1:] != pb[:-1]])
        seg = np.cumsum(np.r_[False, pb[1:] != pb[:-1]])
        size = np.diff(np.r_[starts, pb.shape[0]])
        c = coords[nodes]
        ext = (np.maximum.reduceat(c, starts, axis=0) - np.minimum.reduceat(c, starts, axis=0)) / hk
        axis = np.argmax(ext, axis=1)
        val = c[np.arange(c.shape[0]), axis[seg]]
        o2 = np.lexsort((val, seg))                      # inside each patch: nodes by the chosen coordinate
        rank = np.empty(o2.shape[0], dtype=np.int64)
        rank[o2] = np.arange(o2.shape[0]) - starts[seg[o2]]
        upper = rank >= (size[seg] + 1) // 2
        if (size < 2).any():
            raise ValueError("a single node touches more elements than one patch may hold")
        patch[nodes[upper]] = npat + seg[upper]
    raise RuntimeError("partition_nodes did not converge")


def partition_nodes_device(n_nodes, conn_d, max_elems=64, as_host=True, tile=None):
    """partition_nodes for meshes numbered like a structured grid, on the device (torch integer plumbing, the same
    arithmetic: lattice test, grid binning with patch ids in grid order).  Returns the int64 [N] map as a HOST array (as a
    device tensor with as_host=False), or
    None when the mesh is not lattice-numbered or some patch would touch more than max_elems elements (the NumPy
    version with its bisection pass then takes over).  On lattice meshes the two agree exactly (tests/test_gpu_fused.py)."""
    import torch
    conn = conn_d.to(torch.int64)
    E, N = int(conn.shape[0]), int(n_nodes)
    if E < 4:
        return None
    d01, d32 = (conn[:, 1] - conn[:, 0]).abs(), (conn[:, 2] - conn[:, 3]).abs()
    d03, d12 = (conn[:, 3] - conn[:, 0]).abs(), (conn[:, 2] - conn[:, 1]).abs()
    lo, hi = torch.minimum(d01, d03), torch.maximum(d01, d03)
    vals, cnts = torch.unique(hi, return_counts=True)
    L = int(vals[int(torch.argmax(cnts))])
    if L < 2:
        return None
    ok = (lo == 1) & (hi == L) & (d32 == d01) & (d12 == d03)
    if float(ok.double().mean()) < 0.9:
        return None
    best = (1, 1)
    for a in range(1, GRID_NODES + 1):
        for bb in range(1, 40):
            if (a + 1) * (bb + 1) <= max_elems and a * bb > best[0] * best[1]:
                best = (a, bb)
    used = torch.zeros(N, dtype=torch.bool, device=conn.device)
    used[conn.reshape(-1)] = True
    idx = torch.arange(N, dtype=torch.int64, device=conn.device)
    li, lj = idx % L, idx // L
    # lattice coordinates are integers: floor((c - (min - 0.5)) / side) == (c - min) // side for integer c, side
    umin_i, umin_j = int(li[used].min()), int(lj[used].min())
    bi = torch.where(used, (li - umin_i) // best[0], torch.zeros_like(li))
    bj = torch.where(used, (lj - umin_j) // best[1], torch.zeros_like(lj))
    nbi, nbj = int(bi[used].max()) + 1, int(bj[used].max()) + 1
    bi, bj = torch.clamp(bi, max=nbi - 1), torch.clamp(bj, max=nbj - 1)
    key = _tiled_patch_key(bi, bj, nbi, tile)
    _, patch = torch.unique(key, return_inverse=True)
    npat = int(patch.max()) + 1
    pc, _ = torch.sort(patch[conn], dim=1)
    first = torch.ones_like(pc, dtype=torch.bool)
    first[:, 1:] = pc[:, 1:] != pc[:, :-1]
    cnt = torch.bincount(pc[first], minlength=npat)
    if int(cnt.max()) > max_elems:
        return None
    return patch.cpu().numpy() if as_host else patch


class FusedQuadPlan:
    """Host arrays of fs_fused_assemble_quad's plan: pat_eptr, pat_elems, pat_rptr, recs (one 64-byte record per lane),
    built by the native fs_fused_plan_build (csrc/femshell_fused_plan.cpp; the layout rules are documented at
    fs_fused_rec in include/femshell_hip.h):

    A lane evaluates TWO contributions (step 0, step 1).  A block with n contributions (element order x0, x1, ...):
      n = 3, 4   two adjacent lanes A, B:  A -> x0, x1;  B -> x2, x3 and, chained to A, ((x0 + x1) + x2) + x3
      n = 2      one lane: x0 + x1                          (n < 4: + 0.0, the padded sum of fs_csr_numeric_desc)
      n <= 1     half a lane: two such blocks share a lane (step 0 -> block 0, step 1 -> block 1)
      n > 4      ceil(n / 2) chained lanes inside one 16-lane row, summed 0 + x0 + x1 + ... like fs_csr_numeric_desc
    """

    def __init__(self, plan, conn, coords, max_patch_elems=64, node_patch=None):
        """plan: symbolic.BlockPlan (host) of a uniform quad mesh with d = 6: its node graph (nbr_ptr, nbr) is the CSR
        pattern the kernel writes into, ghost columns of distributed plans included."""
        import ctypes
        from . import _lib
        conn = np.ascontiguousarray(conn, dtype=np.int64)
        if plan.d != 6 or conn.ndim != 2 or conn.shape[1] != 4:
            raise ValueError("the fused kernel is built for uniform MITC4+ meshes (4 nodes, 6 dofs per node)")
        if not 16 <= max_patch_elems <= 64:
            raise ValueError("max_patch_elems must lie in [16, 64]")
        self.E, self.max_patch_elems = int(conn.shape[0]), int(max_patch_elems)
        if node_patch is None:
            node_patch = partition_nodes(coords, conn, max_patch_elems)
        node_patch = np.ascontiguousarray(node_patch, dtype=np.int64)
        nbr_ptr = np.ascontiguousarray(plan.nbr_ptr, dtype=np.int64)
        nbr = np.ascontiguousarray(plan.nbr, dtype=np.int32)
        lib = _lib.load()
        h = ctypes.c_void_p()
        npat, nslots, nrecs, nev = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.fs_fused_plan_build(conn.ctypes.data, self.E, int(plan.n_nodes), nbr_ptr.ctypes.data, nbr.ctypes.data,
                                           node_patch.ctypes.data, self.max_patch_elems, ctypes.byref(h), ctypes.byref(npat),
                                           ctypes.byref(nslots), ctypes.byref(nrecs)), "fs_fused_plan_build")
        try:
            self.n_patches, self.n_elem_slots = int(npat.value), int(nslots.value)
            self.pat_eptr = np.empty(self.n_patches + 1, dtype=np.int64)
            self.pat_rptr = np.empty(self.n_patches + 1, dtype=np.int64)
            self.pat_elems = np.empty(self.n_elem_slots, dtype=np.int32)
            self.recs = np.empty(int(nrecs.value), dtype=REC_DTYPE)
            _lib.check(lib.fs_fused_plan_export(h, self.pat_eptr.ctypes.data, self.pat_elems.ctypes.data,
                                                self.pat_rptr.ctypes.data, self.recs.ctypes.data, ctypes.byref(nev)),
                       "fs_fused_plan_export")
            self.n_evals = int(nev.value)
        finally:
            lib.fs_fused_plan_free(h)

    def stats(self):
        """Recompute factor of stage 1, lane fill of stage 2 (diagnostics for DESIGN.md / bench.py)."""
        rounds = (np.diff(self.pat_rptr) + 255) // 256
        tiles = (np.diff(self.pat_eptr) + 15) // 16
        return {"patches": self.n_patches, "elements_per_patch": self.n_elem_slots / max(self.n_patches, 1),
                "stage1_recompute": self.n_elem_slots / max(self.E, 1),
                "stage1_tile_fill": self.n_elem_slots / max(int(tiles.sum()) * 16, 1),
                "stage2_lane_fill": self.n_evals / max(int(rounds.sum()) * 512, 1),
                "stage2_wave_fill": self.n_evals / max(2 * self.recs.shape[0], 1),
                "record_bytes_per_element": self.recs.nbytes / max(self.E, 1)}

    def to_device(self, device):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        return {"pat_eptr": t(self.pat_eptr), "pat_elems": t(self.pat_elems), "pat_rptr": t(self.pat_rptr),
                "recs": t(self.recs.view(np.uint8).reshape(-1))}


class FusedQuadPlanDevice:
    """The same plan built on the device (fs_fused_plan_device_begin / _finish, csrc/femshell_fused_plan_dev.hip): the
    four arrays never exist on the host.  plan: symbolic.DeviceBlockPlan (device nbr_ptr / nbr), conn_d: device int64
    [E][4], node_patch_d: device int64 [N] (partition_nodes_device(..., as_host=False), or an uploaded
    partition_nodes).  Raises FemShellHipError like FusedQuadPlan; "device plan capacity exceeded" means: use the host
    builder for this mesh."""

    def __init__(self, plan, conn_d, node_patch_d, max_patch_elems=64):
        import ctypes
        import torch
        from . import _lib
        if plan.d != 6 or conn_d.ndim != 2 or conn_d.shape[1] != 4 or conn_d.dtype != torch.int64:
            raise ValueError("the fused kernel is built for uniform MITC4+ meshes (4 nodes, 6 dofs per node)")
        if not 16 <= max_patch_elems <= 64:
            raise ValueError("max_patch_elems must lie in [16, 64]")
        dev = conn_d.device
        conn_d = conn_d.contiguous()
        node_patch_d = node_patch_d.to(device=dev, dtype=torch.int64).contiguous()
        nbr_ptr, nbr = plan.nbr_ptr.contiguous(), plan.nbr.contiguous()
        if nbr_ptr.dtype != torch.int64 or nbr.dtype != torch.int32 or nbr_ptr.numel() != int(plan.n_nodes) + 1:
            raise ValueError("the device plan builder needs the node graph of a DeviceBlockPlan (int64 nbr_ptr, int32 nbr)")
        if node_patch_d.numel() != int(plan.n_nodes):
            raise ValueError("node_patch must hold one patch id per node of the plan")
        self.E, self.max_patch_elems = int(conn_d.shape[0]), int(max_patch_elems)
        lib = _lib.load()
        h = ctypes.c_void_p()
        npat, nslots, nrecs, nev = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.fs_fused_plan_device_begin(_lib.ptr(conn_d), self.E, int(plan.n_nodes), _lib.ptr(nbr_ptr), _lib.ptr(nbr),
                                                  _lib.ptr(node_patch_d), self.max_patch_elems, ctypes.byref(h),
                                                  ctypes.byref(npat), ctypes.byref(nslots), ctypes.byref(nrecs),
                                                  ctypes.byref(nev), _lib.stream_ptr()), "fs_fused_plan_device_begin")
        self.n_patches, self.n_elem_slots, self.n_recs, self.n_evals = (int(npat.value), int(nslots.value),
                                                                        int(nrecs.value), int(nev.value))
        try:
            self._d = {"pat_eptr": torch.empty(self.n_patches + 1, dtype=torch.int64, device=dev),
                       "pat_elems": torch.empty(self.n_elem_slots, dtype=torch.int32, device=dev),
                       "pat_rptr": torch.empty(self.n_patches + 1, dtype=torch.int64, device=dev),
                       "recs": torch.empty(self.n_recs * REC_DTYPE.itemsize, dtype=torch.uint8, device=dev)}
        except BaseException:
            lib.fs_fused_plan_device_free(h)
            raise
        d = self._d
        _lib.check(lib.fs_fused_plan_device_finish(h, _lib.ptr(d["pat_eptr"]), _lib.ptr(d["pat_elems"]), _lib.ptr(d["pat_rptr"]),
                                                   _lib.ptr(d["recs"]), _lib.stream_ptr()), "fs_fused_plan_device_finish")

    def stats(self):
        import torch
        d = self._d
        rounds = int(((torch.diff(d["pat_rptr"]) + 255) // 256).sum())
        tiles = int(((torch.diff(d["pat_eptr"]) + 15) // 16).sum())
        return {"patches": self.n_patches, "elements_per_patch": self.n_elem_slots / max(self.n_patches, 1),
                "stage1_recompute": self.n_elem_slots / max(self.E, 1),
                "stage1_tile_fill": self.n_elem_slots / max(tiles * 16, 1),
                "stage2_lane_fill": self.n_evals / max(rounds * 512, 1),
                "stage2_wave_fill": self.n_evals / max(2 * self.n_recs, 1),
                "record_bytes_per_element": self.n_recs * REC_DTYPE.itemsize / max(self.E, 1)}

    def to_device(self, device):
        return self._d

    def to_host(self):
        """The four arrays as NumPy (tests, diagnostics)."""
        d = self._d
        return {"pat_eptr": d["pat_eptr"].cpu().numpy(), "pat_elems": d["pat_elems"].cpu().numpy(),
                "pat_rptr": d["pat_rptr"].cpu().numpy(), "recs": d["recs"].cpu().numpy().view(REC_DTYPE)}

    # host views with FusedQuadPlan's names, copied from the device when somebody asks (tests, diagnostics)
    pat_eptr = property(lambda self: self._d["pat_eptr"].cpu().numpy())
    pat_elems = property(lambda self: self._d["pat_elems"].cpu().numpy())
    pat_rptr = property(lambda self: self._d["pat_rptr"].cpu().numpy())
    recs = property(lambda self: self._d["recs"].cpu().numpy().view(REC_DTYPE))


# ---------------------------------------------------------------------------------------------------------------------
# HEXA8 (fs_fused_assemble_hex8)
# ---------------------------------------------------------------------------------------------------------------------
HEX_NODE_DTYPE = np.dtype([("sbase", "i4"), ("m", "i4")])
HEX_RUN_DTYPE = np.dtype([("gdst", "i8"), ("sbase", "i4"), ("len", "i4")])
HEX_CELL = (4, 4, 3)       # nodes per patch along the numbering direction, the next one, the slowest one: 48 nodes touch
                           # 5 x 5 x 4 = 100 elements (FS_HEX_MAX_PATCH_ELEMS: the LDS image of one workgroup)


def _bisect_patches(patch, coords, hk, used, bad, npat):
    """One round of median bisection, in place: every patch flagged in `bad` is cut in two along its longest extent
    (measured in node spacings hk); the upper halves get new ids npat, npat + 1, ..."""
    nodes = np.flatnonzero(bad[patch] & used)
    pb = patch[nodes]
    o = np.argsort(pb, kind="stable")
    nodes, pb = nodes[o], pb[o]
    starts = np.flatnonzero(np.r_[True, pb[1:] != pb[:-1]])
    seg = np.cumsum(np.r_[False, pb[1:] != pb[:-1]])
    size = np.diff(np.r_[starts, pb.shape[0]])
    c = coords[nodes]
    ext = (np.maximum.reduceat(c, starts, axis=0) - np.minimum.reduceat(c, starts, axis=0)) / hk
    axis = np.argmax(ext, axis=1)
    val = c[np.arange(c.shape[0]), axis[seg]]
    o2 = np.lexsort((val, seg))
    rank = np.empty(o2.shape[0], dtype=np.int64)
    rank[o2] = np.arange(o2.shape[0]) - starts[seg[o2]]
    upper = rank >= (size[seg] + 1) // 2
    if (size < 2).any():
        raise ValueError("a single node touches more elements than one patch may hold")
    patch[nodes[upper]] = npat + seg[upper]


def partition_nodes_hex(coords, conn, max_elems=100, max_nodes=48, max_lanes=512):
    """node -> patch id for the fused HEXA8 kernel: bins of HEX_CELL nodes on the coordinates (spacing = mean element
    extent per axis; the axis along which consecutive node indices run is the fastest one, so the CSR slabs of a patch
    form long runs and consecutive patches are neighbours), then median bisection of the patches that touch more than
    `max_elems` elements, hold more than `max_nodes` nodes or would need more than `max_lanes` lanes (9 lanes per
    interior node of a structured block: 72 two-step slots for its 64 contributions)."""
    coords = np.asarray(coords, dtype=np.float64)
    conn = np.asarray(conn, dtype=np.int64)
    N, E = coords.shape[0], conn.shape[0]
    if E == 0:
        return np.zeros(N, dtype=np.int64)
    used = np.zeros(N, dtype=bool)
    used[conn.reshape(-1)] = True
    x = coords[conn]
    hk = np.maximum((x.max(axis=1) - x.min(axis=1)).mean(axis=0), 1e-300)
    hk = np.maximum(hk, 1e-3 * float(hk.max()))
    # order of the axes in the node numbering: along which global axis does the node index grow by the smallest step?
    # (structured blocks: 1 along the fastest axis, one row of nodes along the next, one layer along the slowest)
    ea = np.array([0, 3, 4, 7, 0, 1, 4, 5, 0, 1, 2, 3])
    eb = np.array([1, 2, 5, 6, 3, 2, 7, 6, 4, 5, 6, 7])
    sub = conn[:: max(1, E // 4096)]
    dx = np.abs(coords[sub[:, eb]] - coords[sub[:, ea]]) / hk               # [e, 12 edges, 3]
    ax_of = np.argmax(dx, axis=2)
    di = np.abs(sub[:, eb] - sub[:, ea]).astype(np.float64)
    med = [np.median(di[ax_of == k]) if (ax_of == k).any() else np.inf for k in range(3)]
    order_axes = [int(k) for k in np.argsort(med, kind="stable")]
    # node spacing per axis = median edge length along it (the element extents above are inflated by distortion: bins of
    # "4 spacings" would then catch a fifth node now and then and be bisected)
    raw = np.abs(coords[sub[:, eb]] - coords[sub[:, ea]])
    hk = np.array([np.median(raw[:, :, k][ax_of == k]) if (ax_of == k).any() else hk[k] for k in range(3)])
    hk = np.maximum(hk, 1e-3 * float(hk.max()))
    cell = np.empty(3)
    for k, ax in enumerate(order_axes):
        cell[ax] = HEX_CELL[k]
    side = cell * hk
    lo = coords[used].min(axis=0) - 0.5 * hk
    cc = np.where(used[:, None], coords, lo)
    b = np.clip(np.floor((cc - lo) / side).astype(np.int64), 0, None)
    nb = b[used].max(axis=0) + 1
    b = np.minimum(b, nb - 1)
    key = np.zeros(N, dtype=np.int64)
    for ax in reversed(order_axes):
        key = key * nb[ax] + b[:, ax]
    _, patch = np.unique(key, return_inverse=True)
    patch = patch.astype(np.int64)
    valence = np.bincount(conn.reshape(-1), minlength=N)
    for _ in range(64):
        npat = int(patch.max()) + 1
        cnt = _elements_per_patch(patch, conn, npat)
        nn = np.bincount(patch[used], minlength=npat)
        lanes = np.bincount(patch[used], weights=np.ceil(np.maximum(valence[used], 1) * 1.125), minlength=npat)
        bad = (cnt > max_elems) | (nn > max_nodes) | (lanes > max_lanes)
        if not bad.any():
            return patch
        _bisect_patches(patch, coords, hk, used, bad, npat)
    raise RuntimeError("partition_nodes_hex did not converge")


def partition_nodes_hex_device(n_nodes, conn_d, max_elems=100, max_nodes=48, max_lanes=512):
    """partition_nodes_hex for brick meshes numbered like a structured block (node index = (k * Nj + j) * Ni + i: the
    element edges join indices that differ by 1, Ni and Ni * Nj, whatever the geometry), on the device (torch integer
    plumbing): cells of HEX_CELL nodes on the (i, j, k) lattice, patch ids in grid order.  Exactly 4 x 4 x 3 nodes per
    interior patch (the coordinate bins of the NumPy version catch a fifth node now and then on jittered meshes and are
    bisected: 45 nodes per patch on configs[4]'s block).  Returns the int64 [N] map as a HOST array, or None when the
    mesh is not numbered that way or a patch would exceed a limit (the NumPy version then takes over)."""
    import torch
    conn = conn_d.to(torch.int64)
    E, N = int(conn.shape[0]), int(n_nodes)
    if E < 8 or conn.shape[1] != 8:
        return None
    c0 = conn[:, 0]
    d = torch.stack([(conn[:, 1] - c0).abs(), (conn[:, 3] - c0).abs(), (conn[:, 4] - c0).abs()], dim=1)
    d, _ = torch.sort(d, dim=1)

    def mode(v):
        vals, cnts = torch.unique(v, return_counts=True)
        return int(vals[int(torch.argmax(cnts))])
    L1, L2 = mode(d[:, 1]), mode(d[:, 2])
    if L1 < 2 or L2 < 2 * L1 or L2 % L1:
        return None
    # all 12 edges of an element: 4 along each lattice direction
    ea = torch.tensor([0, 3, 4, 7, 0, 1, 4, 5, 0, 1, 2, 3], device=conn.device)
    eb = torch.tensor([1, 2, 5, 6, 3, 2, 7, 6, 4, 5, 6, 7], device=conn.device)
    de, _ = torch.sort((conn[:, eb] - conn[:, ea]).abs(), dim=1)
    want = torch.tensor([1] * 4 + [L1] * 4 + [L2] * 4, device=conn.device)
    if float((de == want).all(dim=1).double().mean()) < 0.9:
        return None
    used = torch.zeros(N, dtype=torch.bool, device=conn.device)
    used[conn.reshape(-1)] = True
    idx = torch.arange(N, dtype=torch.int64, device=conn.device)
    lat = (idx % L1, (idx // L1) % (L2 // L1), idx // L2)
    key, mult = torch.zeros_like(idx), 1
    for ax in range(3):
        lo = int(lat[ax][used].min())
        b = torch.where(used, (lat[ax] - lo) // HEX_CELL[ax], torch.zeros_like(idx))
        key = key + b * mult
        mult *= int(b.max()) + 1
    _, patch = torch.unique(key, return_inverse=True)
    npat = int(patch.max()) + 1
    pc, _ = torch.sort(patch[conn], dim=1)
    first = torch.ones_like(pc, dtype=torch.bool)
    first[:, 1:] = pc[:, 1:] != pc[:, :-1]
    cnt = torch.bincount(pc[first], minlength=npat)
    nn = torch.bincount(patch[used], minlength=npat)
    valence = torch.bincount(conn.reshape(-1), minlength=N)
    lanes = torch.bincount(patch[used], weights=torch.ceil(valence[used].clamp(min=1).double() * 1.125), minlength=npat)
    if int(cnt.max()) > max_elems or int(nn.max()) > max_nodes or float(lanes.max()) > max_lanes:
        return None
    return patch.cpu().numpy()


class FusedHexPlan:
    """Host arrays of fs_fused_assemble_hex8's plan (layout: include/femshell_hip.h), built by the native
    fs_fused_hex_plan_build (csrc/femshell_fused_plan.cpp)."""

    def __init__(self, plan, conn, coords, node_patch=None):
        import ctypes
        from . import _lib
        conn = np.ascontiguousarray(conn, dtype=np.int64)
        if plan.d != 3 or conn.ndim != 2 or conn.shape[1] != 8:
            raise ValueError("the fused HEXA8 kernel is built for uniform 8-node brick meshes (3 dofs per node)")
        self.E = int(conn.shape[0])
        if node_patch is None:
            node_patch = partition_nodes_hex(coords, conn)
        node_patch = np.ascontiguousarray(node_patch, dtype=np.int64)
        nbr_ptr = np.ascontiguousarray(plan.nbr_ptr, dtype=np.int64)
        nbr = np.ascontiguousarray(plan.nbr, dtype=np.int32)
        lib = _lib.load()
        h = ctypes.c_void_p()
        c = [ctypes.c_int64() for _ in range(5)]
        _lib.check(lib.fs_fused_hex_plan_build(conn.ctypes.data, self.E, int(plan.n_nodes), nbr_ptr.ctypes.data, nbr.ctypes.data,
                                               node_patch.ctypes.data, ctypes.byref(h), *[ctypes.byref(v) for v in c]),
                   "fs_fused_hex_plan_build")
        try:
            npat, nslots, nnodes, nruns, nlanes = (int(v.value) for v in c)
            self.n_patches, self.n_elem_slots, self.n_lanes = npat, nslots, nlanes
            self.pat_eptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_nptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_rptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_lptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_elems = np.empty(nslots, dtype=np.int32)
            self.nodes = np.empty(nnodes, dtype=HEX_NODE_DTYPE)
            self.runs = np.empty(nruns, dtype=HEX_RUN_DTYPE)
            self.steps = np.empty((nlanes, 8), dtype=np.uint32)
            nev = ctypes.c_int64()
            _lib.check(lib.fs_fused_hex_plan_export(h, self.pat_eptr.ctypes.data, self.pat_elems.ctypes.data,
                                                    self.pat_nptr.ctypes.data, self.nodes.ctypes.data, self.pat_rptr.ctypes.data,
                                                    self.runs.ctypes.data, self.pat_lptr.ctypes.data, self.steps.ctypes.data,
                                                    ctypes.byref(nev)), "fs_fused_hex_plan_export")
            self.n_evals = int(nev.value)
        finally:
            lib.fs_fused_hex_plan_free(h)

    def stats(self):
        return {"patches": self.n_patches, "elements_per_patch": self.n_elem_slots / max(self.n_patches, 1),
                "stage1_recompute": self.n_elem_slots / max(self.E, 1),
                "nodes_per_patch": self.nodes.shape[0] / max(self.n_patches, 1),
                "runs_per_patch": self.runs.shape[0] / max(self.n_patches, 1),
                "step_fill": self.n_evals / max(self.steps.size, 1),
                "record_bytes_per_element": (self.steps.nbytes + self.runs.nbytes + self.nodes.nbytes) / max(self.E, 1)}

    def to_device(self, device):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(device)
        return {k: t(getattr(self, k)) for k in ("pat_eptr", "pat_elems", "pat_nptr", "nodes", "pat_rptr", "runs",
                                                 "pat_lptr", "steps")}


# ---------------------------------------------------------------------------------------------------------------------
# MITC3+ (fs_fused_assemble_tri)
# ---------------------------------------------------------------------------------------------------------------------
TRI_NODE_DTYPE = np.dtype([("gbase", "i8"), ("m", "i4"), ("pad", "i4")])
TRI_CELL = (7, 10)         # nodes per patch along the numbering direction and across it: 70 nodes touch 2 x 8 x 11 = 176
                           # triangles of a structured triangulation (FS_TRI_MAX_PATCH_ELEMS: the LDS image of a workgroup)


def _tri_lattice(conn, N):
    """(i, j) lattice indices of the nodes when the triangles are the halves of a structured grid's cells (node index =
    j * L + i: the edges of a triangle join indices that differ by 1, L and L +- 1), else None."""
    if conn.shape[0] < 4:
        return None
    c0, c1, c2 = conn[:, 0], conn[:, 1], conn[:, 2]
    d = np.sort(np.abs(np.stack([c1 - c0, c2 - c1, c2 - c0], axis=1)), axis=1)

    def mode(v):
        vals, cnts = np.unique(v, return_counts=True)
        return int(vals[np.argmax(cnts)])
    a, b = mode(d[:, 1]), mode(d[:, 2])
    if mode(d[:, 0]) != 1 or b != a + 1 or a < 2:
        return None
    if ((d[:, 0] == 1) & (d[:, 1] == a) & (d[:, 2] == b)).mean() < 0.9:
        return None
    cand = [L for L in (a, b) if N % L == 0]
    if len(cand) != 1:
        return None                      # diagonal direction not decidable from the indices alone
    L = cand[0]
    idx = np.arange(N, dtype=np.int64)
    return np.stack([idx % L, idx // L], axis=1).astype(np.float64)


def partition_nodes_tri(coords, conn, max_elems=176, max_nodes=96, max_lanes=256):
    """node -> patch id for the fused MITC3+ kernel: cells of TRI_CELL nodes on the lattice of the node numbering when the
    mesh is a triangulated structured grid (curved shells included), else on the coordinates (node spacing = median
    edge length per axis), then median bisection of the patches beyond a limit (elements touched, nodes, lanes: 3 lanes
    of 8 steps per interior node -- 6 + 6 x 2 contributions)."""
    coords = np.asarray(coords, dtype=np.float64)
    conn = np.asarray(conn, dtype=np.int64)
    N, E = coords.shape[0], conn.shape[0]
    if E == 0:
        return np.zeros(N, dtype=np.int64)
    used = np.zeros(N, dtype=bool)
    used[conn.reshape(-1)] = True
    lat = _tri_lattice(conn, N)
    if lat is not None:
        coords, hk, ax = lat, np.ones(2), 0
    else:
        x = coords[conn]
        edges = np.abs(np.concatenate([x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 0] - x[:, 2]]))
        hk = np.maximum(edges.mean(axis=0), 1e-300)
        hk = np.maximum(hk, 1e-3 * float(hk.max()))
        ax = _numbering_axis(coords, used)
    nd = coords.shape[1]
    cell = np.full(nd, float(TRI_CELL[1]))
    cell[ax] = float(TRI_CELL[0])
    if nd == 3:
        cell[:] = 8.0          # a surface in space: cubes of 8 spacings (~64 nodes), cut by the bisection pass where needed
    side = cell * hk
    lo = coords[used].min(axis=0) - 0.5 * hk
    cc = np.where(used[:, None], coords, lo)
    b = np.clip(np.floor((cc - lo) / side).astype(np.int64), 0, None)
    nb = b[used].max(axis=0) + 1
    b = np.minimum(b, nb - 1)
    # (row by row: the 4 x 8 tiles of the quad plan make this kernel, whose CUs hold several workgroups, 3 % slower)
    key = np.zeros(N, dtype=np.int64)
    for k in reversed([ax] + [k for k in range(nd) if k != ax]):
        key = key * nb[k] + b[:, k]
    _, patch = np.unique(key, return_inverse=True)
    patch = patch.astype(np.int64)
    valence = np.bincount(conn.reshape(-1), minlength=N)
    for _ in range(64):
        npat = int(patch.max()) + 1
        cnt = _elements_per_patch(patch, conn, npat)
        nn = np.bincount(patch[used], minlength=npat)
        lanes = np.bincount(patch[used], weights=np.ceil(np.maximum(valence[used], 1) * 0.375 + 0.75), minlength=npat)
        bad = (cnt > max_elems) | (nn > max_nodes) | (lanes > max_lanes)
        if not bad.any():
            return patch
        _bisect_patches(patch, coords, hk, used, bad, npat)
    raise RuntimeError("partition_nodes_tri did not converge")


class FusedTriPlan:
    """Host arrays of fs_fused_assemble_tri's plan (layout: include/femshell_hip.h), built by the native
    fs_fused_tri_plan_build (csrc/femshell_fused_plan.cpp)."""

    def __init__(self, plan, conn, coords, node_patch=None):
        import ctypes
        from . import _lib
        conn = np.ascontiguousarray(conn, dtype=np.int64)
        if plan.d != 6 or conn.ndim != 2 or conn.shape[1] != 3:
            raise ValueError("the fused MITC3+ kernel is built for uniform 3-node shell meshes (6 dofs per node)")
        self.E = int(conn.shape[0])
        if node_patch is None:
            node_patch = partition_nodes_tri(coords, conn)
        node_patch = np.ascontiguousarray(node_patch, dtype=np.int64)
        nbr_ptr = np.ascontiguousarray(plan.nbr_ptr, dtype=np.int64)
        nbr = np.ascontiguousarray(plan.nbr, dtype=np.int32)
        lib = _lib.load()
        h = ctypes.c_void_p()
        c = [ctypes.c_int64() for _ in range(4)]
        _lib.check(lib.fs_fused_tri_plan_build(conn.ctypes.data, self.E, int(plan.n_nodes), nbr_ptr.ctypes.data, nbr.ctypes.data,
                                               node_patch.ctypes.data, ctypes.byref(h), *[ctypes.byref(v) for v in c]),
                   "fs_fused_tri_plan_build")
        try:
            npat, nslots, nnodes, nlanes = (int(v.value) for v in c)
            self.n_patches, self.n_elem_slots, self.n_lanes = npat, nslots, nlanes
            self.pat_eptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_nptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_lptr = np.empty(npat + 1, dtype=np.int64)
            self.pat_elems = np.empty(nslots, dtype=np.int32)
            self.nodes = np.empty(nnodes, dtype=TRI_NODE_DTYPE)
            self.steps = np.empty((nlanes, 8), dtype=np.uint32)
            nev = ctypes.c_int64()
            _lib.check(lib.fs_fused_tri_plan_export(h, self.pat_eptr.ctypes.data, self.pat_elems.ctypes.data,
                                                    self.pat_nptr.ctypes.data, self.nodes.ctypes.data,
                                                    self.pat_lptr.ctypes.data, self.steps.ctypes.data, ctypes.byref(nev)),
                       "fs_fused_tri_plan_export")
            self.n_evals = int(nev.value)
        finally:
            lib.fs_fused_tri_plan_free(h)

    def stats(self):
        return {"patches": self.n_patches, "elements_per_patch": self.n_elem_slots / max(self.n_patches, 1),
                "stage1_recompute": self.n_elem_slots / max(self.E, 1),
                "nodes_per_patch": self.nodes.shape[0] / max(self.n_patches, 1),
                "step_fill": self.n_evals / max(self.steps.size, 1),
                "record_bytes_per_element": (self.steps.nbytes + self.nodes.nbytes) / max(self.E, 1)}

    def to_device(self, device):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(device)
        return {k: t(getattr(self, k)) for k in ("pat_eptr", "pat_elems", "pat_nptr", "nodes", "pat_lptr", "steps")}
