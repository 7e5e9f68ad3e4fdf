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
import numpy as np

from .symbolic import _T16_RG

REC_DTYPE = np.dtype([("dst", "i8"), ("dst_t", "i8"), ("stride", "i4"), ("stride_t", "i4"), ("ev", "u4"), ("aux", "u4")])
assert REC_DTYPE.itemsize == 32

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


def partition_nodes(coords, conn, max_elems=64, target_nodes=49):
    """node -> patch id (int64 [N]); the elements touching any patch number <= max_elems.

    Grid binning with cells of sqrt(target_nodes) mean element edges (patch ids follow the grid row-major, so
    consecutive patches are spatial neighbours), then recursive median bisection, along the longest extent, of the
    patches that still touch too many elements."""
    coords = np.asarray(coords, dtype=np.float64)
    conn = np.asarray(conn, dtype=np.int64)
    N, E = coords.shape[0], conn.shape[0]
    if E == 0:
        return np.zeros(N, dtype=np.int64)
    x = coords[conn]
    # mean edge length per direction of the element's local numbering (anisotropic meshes get anisotropic cells
    # only through the bisection step; the grid uses the geometric mean)
    h1 = float(np.linalg.norm(x[:, 1] - x[:, 0], axis=1).mean())
    h2 = float(np.linalg.norm(x[:, 3] - x[:, 0], axis=1).mean())
    h = max(np.sqrt(h1 * h2), 1e-300)
    side = np.sqrt(float(target_nodes)) * h
    used = np.zeros(N, dtype=bool)
    used[conn.reshape(-1)] = True
    lo = coords[used].min(axis=0) - 0.5 * h
    b = np.floor((coords - lo) / side).astype(np.int64)
    b = np.clip(b, 0, None)
    nb = b[used].max(axis=0) + 1
    b = np.minimum(b, nb - 1)
    key = (b[:, 2] * nb[1] + b[:, 1]) * nb[0] + b[:, 0]
    _, patch = np.unique(key, return_inverse=True)
    patch = patch.astype(np.int64)
    earange = np.repeat(np.arange(E, dtype=np.int64), conn.shape[1])
    for _ in range(64):
        npat = int(patch.max()) + 1
        pe = np.unique(patch[conn].reshape(-1) * E + earange)
        cnt = np.bincount(pe // E, minlength=npat)
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
        ext = np.maximum.reduceat(c, starts, axis=0) - np.minimum.reduceat(c, starts, axis=0)
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


class FusedQuadPlan:
    """Host arrays of fs_fused_assemble_quad's plan: pat_eptr, pat_elems, pat_rptr, recs, ser_ev."""

    def __init__(self, plan, conn, coords, max_patch_elems=64, node_patch=None):
        """plan: symbolic.BlockPlan (host) of a uniform quad mesh with d = 6 (tiled or element-major block indices)."""
        conn = np.asarray(conn, dtype=np.int64)
        if plan.d != 6 or conn.ndim != 2 or conn.shape[1] != 4:
            raise ValueError("the fused kernel is built for uniform MITC4+ meshes (4 nodes, 6 dofs per node)")
        if not 16 <= max_patch_elems <= 80:
            raise ValueError("max_patch_elems must lie in [16, 80]")
        E, N, d = conn.shape[0], plan.n_nodes, 6
        self.max_patch_elems = int(max_patch_elems)
        if node_patch is None:
            node_patch = partition_nodes(coords, conn, max_patch_elems)
        node_patch = np.asarray(node_patch, dtype=np.int64)
        # ---- tasks: blocks with I <= J, plus pattern-only blocks whose transposed partner is absent ----
        I, J = plan.blk_row.astype(np.int64), plan.nbr.astype(np.int64)
        nblk = plan.nblk
        key, tkey = I * N + J, J * N + I
        pidx = np.minimum(np.searchsorted(key, tkey), max(nblk - 1, 0))
        has_partner = key[pidx] == tkey if nblk else np.zeros(0, dtype=bool)
        cnt_all = np.diff(plan.blk_ptr)
        if np.any(~has_partner & (cnt_all > 0)):
            raise ValueError("block with contributions but no transposed partner: connectivity is not symmetric")
        m = np.diff(plan.nbr_ptr)
        r0 = plan.nbr_ptr[I]
        dst_all = d * d * r0 + d * (np.arange(nblk, dtype=np.int64) - r0)
        stride_all = (d * m[I]).astype(np.int64)
        tb = np.flatnonzero((I <= J) | ~has_partner)
        t_cnt = cnt_all[tb]
        t_patch_raw = node_patch[I[tb]]
        # compact patch ids to those that own a task, keeping their order
        upatch, t_patch = np.unique(t_patch_raw, return_inverse=True)
        npat = upatch.shape[0]
        serial = t_cnt > 4
        width = np.where(serial, 1, np.where(t_cnt >= 3, 4, np.where(t_cnt == 2, 2, 1))).astype(np.int64)
        cls = np.where(serial, 3, np.where(width == 4, 0, np.where(width == 2, 1, 2)))
        order = np.lexsort((np.arange(tb.shape[0]), cls, t_patch))
        tb, t_cnt, t_patch, serial, width = tb[order], t_cnt[order], t_patch[order], serial[order], width[order]
        # lane offset of every task inside its patch (widths sorted 4, 2, 1: no group straddles a quad)
        wsum = np.cumsum(width) - width
        pstart = np.searchsorted(t_patch, np.arange(npat + 1))
        pbase = wsum[np.minimum(pstart[:-1], max(tb.shape[0] - 1, 0))] if tb.shape[0] else np.zeros(npat, dtype=np.int64)
        lanes = np.zeros(npat, dtype=np.int64)
        np.add.at(lanes, t_patch, width)
        lanes = (lanes + 3) // 4 * 4
        self.pat_rptr = np.zeros(npat + 1, dtype=np.int64)
        np.cumsum(lanes, out=self.pat_rptr[1:])
        base = self.pat_rptr[t_patch] + (wsum - pbase[t_patch])
        nrec = int(self.pat_rptr[-1])
        rec = np.zeros(nrec, dtype=REC_DTYPE)
        rec["dst_t"] = -1
        # ---- contributions -> (element, pair, transposed) ----
        start = plan.blk_ptr[tb]
        tot = int(t_cnt.sum())
        owner = np.repeat(np.arange(tb.shape[0], dtype=np.int64), t_cnt)          # task of every contribution
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(t_cnt) - t_cnt, t_cnt)
        cc = plan.contrib[start[owner] + within]
        src, tr = cc >> 1, (cc & 1).astype(np.uint32)
        if plan.tiled:
            pos = src % 160
            el = (src // 160) * 16 + _T16_Q[pos]
            pr = _T16_P[pos]
        else:
            el, pr = src // 10, src % 10
        if tot and (el.min() < 0 or el.max() >= E):
            raise ValueError("plan refers to elements outside the mesh")
        # ---- patch element lists and slots ----
        pe_key = t_patch[owner] * E + el
        upe, inv = np.unique(pe_key, return_inverse=True)
        self.pat_elems = (upe % E).astype(np.int32)
        self.pat_eptr = np.searchsorted(upe // E, np.arange(npat + 1)).astype(np.int64)
        per_patch = np.diff(self.pat_eptr)
        if per_patch.size and per_patch.max() > max_patch_elems:
            raise ValueError(f"a patch touches {int(per_patch.max())} elements (> {max_patch_elems}): "
                             "node_patch does not respect the limit")
        slot = (inv - self.pat_eptr[t_patch[owner]]).astype(np.uint32)
        code = slot | (_PAIR_I[pr] << 8) | (_PAIR_J[pr] << 10) | (tr << 12)
        # ---- records ----
        ser_c = serial[owner]
        lane_of = base[owner] + within
        nn = np.minimum(t_cnt, 4).astype(np.uint32)
        rec["ev"][lane_of[~ser_c]] = code[~ser_c] | (np.uint32(1) << 17) | (nn[owner[~ser_c]] << 13)
        # leaders (first lane of every task)
        lead_ev = rec["ev"][base] | (np.uint32(1) << 16)
        ns = int(serial.sum())
        self.ser_ev = np.zeros(max(int(t_cnt[serial].sum()), 1), dtype=np.uint32)
        if ns:
            if t_cnt[serial].max() >= 4096:
                raise ValueError("more than 4095 contributions to one block")
            self.ser_ev[:int(ser_c.sum())] = code[ser_c]                   # contributions are grouped by task already
            off = np.cumsum(t_cnt[serial]) - t_cnt[serial]
            aux = np.zeros(tb.shape[0], dtype=np.uint32)
            aux[serial] = off.astype(np.uint32)
            rec["aux"][base] = aux
            lead_ev = np.where(serial, (np.uint32(7) << 13) | (np.uint32(1) << 16) | (t_cnt.astype(np.uint32) << 20), lead_ev)
        rec["ev"][base] = lead_ev
        rec["dst"][base] = dst_all[tb]
        rec["stride"][base] = stride_all[tb]
        off_diag = (I[tb] < J[tb]) & has_partner[tb]
        rec["dst_t"][base] = np.where(off_diag, dst_all[pidx[tb]], -1)
        rec["stride_t"][base] = np.where(off_diag, stride_all[pidx[tb]], 0)
        self.recs = rec
        self.n_patches = npat
        self.n_tasks = int(tb.shape[0])
        self.n_evals = tot
        self.n_elem_slots = int(self.pat_elems.shape[0])
        self.E = E

    def stats(self):
        """Recompute factor of stage 1, lane fill of stage 2 (diagnostics for DESIGN.md / bench.py)."""
        rounds = (np.diff(self.pat_rptr) + 255) // 256
        tiles = (np.diff(self.pat_eptr) + 15) // 16
        return {"patches": self.n_patches, "elements_per_patch": self.n_elem_slots / max(self.n_patches, 1),
                "stage1_recompute": self.n_elem_slots / max(self.E, 1),
                "stage1_tile_fill": self.n_elem_slots / max(int(tiles.sum()) * 16, 1),
                "stage2_lane_fill": self.n_evals / max(int(rounds.sum()) * 256, 1),
                "record_bytes_per_element": self.recs.nbytes / max(self.E, 1)}

    def to_device(self, device):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        return {"pat_eptr": t(self.pat_eptr), "pat_elems": t(self.pat_elems), "pat_rptr": t(self.pat_rptr),
                "recs": t(self.recs.view(np.uint8).reshape(-1)), "ser_ev": t(self.ser_ev.view(np.int32))}
