"""NumPy statement of the fused kernel's patch plan (the round-2 prototype of csrc/femshell_fused_plan.cpp): test
reference only -- tests/test_fused_plan_host.py checks that the native builder produces the same arrays bit for bit."""
import numpy as np

from fem_shell_amd.fused import (FL_CHAIN, FL_MAXCHAIN_SHIFT, FL_OUT0, FL_OUT1, FL_PAD, FL_POS_SHIFT, FL_ZERO_START,
                                 GROUP_NODES, REC_DTYPE, _PAIR_I, _PAIR_J, _T16_P, _T16_Q, partition_nodes)


class NumpyFusedQuadPlan:
    """Host arrays of fs_fused_assemble_quad's plan: pat_eptr, pat_elems, pat_rptr, recs (one 64-byte record per lane).

    A lane evaluates TWO contributions (step 0, step 1).  A block with n contributions (element order x0, x1, ...):
      n = 3, 4   two adjacent lanes A, B:  A -> x0, x1;  B -> x2, x3 and, chained to A, ((x0 + x1) + x2) + x3
      n = 2      one lane: x0 + x1                          (n < 4: + 0.0, the padded sum of fs_csr_numeric_desc)
      n <= 1     half a lane: two such blocks share a lane (step 0 -> block 0, step 1 -> block 1)
      n > 4      ceil(n / 2) chained lanes inside one 16-lane row, summed 0 + x0 + x1 + ... like fs_csr_numeric_desc
    """

    def __init__(self, plan, conn, coords, max_patch_elems=64, node_patch=None):
        """plan: symbolic.BlockPlan (host) of a uniform quad mesh with d = 6 (tiled or element-major block indices)."""
        conn = np.asarray(conn, dtype=np.int64)
        if plan.d != 6 or conn.ndim != 2 or conn.shape[1] != 4:
            raise ValueError("the fused kernel is built for uniform MITC4+ meshes (4 nodes, 6 dofs per node)")
        if not 16 <= max_patch_elems <= 64:
            raise ValueError("max_patch_elems must lie in [16, 64]")
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
        if cnt_all.size and cnt_all.max() > 32:
            raise ValueError("a node pair shared by more than 32 elements is beyond the fused kernel (use fused=False)")
        m = np.diff(plan.nbr_ptr)
        r0 = plan.nbr_ptr[I]
        dst_all = d * d * r0 + d * (np.arange(nblk, dtype=np.int64) - r0)
        stride_all = (d * m[I]).astype(np.int64)
        tb = np.flatnonzero((I <= J) | ~has_partner)
        nt = tb.shape[0]
        t_cnt = cnt_all[tb]
        lead = I[tb]
        # compact patch ids to those that own a task, keeping their order
        upatch, t_patch = np.unique(node_patch[lead], return_inverse=True)
        npat = upatch.shape[0]
        # ---- wave groups: the tasks led by GROUP_NODES consecutive (by index) nodes of a patch share a wave; blocks
        # with more than 4 contributions get 16-lane rows of their own in extra groups at the end of the patch ----
        unk, ninv = np.unique(t_patch * N + lead, return_inverse=True)          # leading nodes by (patch, node)
        npatch_of = unk // N
        nfirst = np.searchsorted(npatch_of, np.arange(npat + 1))
        rank = np.arange(unk.shape[0], dtype=np.int64) - nfirst[npatch_of]
        long_t = t_cnt > 4
        nlong = np.bincount(t_patch[long_t], minlength=npat)
        gnorm = (np.diff(nfirst) + GROUP_NODES - 1) // GROUP_NODES             # normal groups per patch
        gpp = gnorm + (nlong + 3) // 4
        gbase = np.cumsum(gpp) - gpp
        t_group = (gbase[npatch_of] + rank // GROUP_NODES)[ninv]
        if long_t.any():
            lt = np.flatnonzero(long_t)
            o = np.lexsort((J[tb][lt], lead[lt], t_patch[lt]))
            lt = lt[o]
            lp = t_patch[lt]
            lrank = np.arange(lt.shape[0]) - np.searchsorted(lp, lp)
            t_group[lt] = gbase[lp] + gnorm[lp] + lrank // 4
            long_row = np.zeros(nt, dtype=np.int64)
            long_row[lt] = lrank % 4
        ngroups = int(gpp.sum())
        # half-lanes a task needs: 4 (n = 3, 4), 2 (n = 2), 1 (n <= 1); long tasks: a 16-lane row = 32 half-lanes
        hw = np.where(long_t, 32, np.where(t_cnt >= 3, 4, np.where(t_cnt == 2, 2, 1))).astype(np.int64)
        cls = np.where(long_t, 0, np.where(hw == 4, 1, np.where(hw == 2, 2, 3)))
        order = np.lexsort((J[tb], lead, cls, t_group))
        tb, t_cnt, t_patch, t_group, hw, long_t, lead = (a[order] for a in (tb, t_cnt, t_patch, t_group, hw, long_t, lead))
        hsum = np.cumsum(hw) - hw
        gstart = np.searchsorted(t_group, np.arange(ngroups + 1))
        ghalf = np.zeros(ngroups, dtype=np.int64)
        np.add.at(ghalf, t_group, hw)
        gpad = ((ghalf + 1) // 2 + 63) // 64 * 64                              # lanes of a group: whole waves
        grec = np.cumsum(gpad) - gpad
        hbase = 2 * grec[t_group] + (hsum - hsum[gstart[:-1]][t_group])        # first half-lane of every task
        nrec = int(gpad.sum())
        self.pat_rptr = (np.append(grec[np.minimum(gbase, max(ngroups - 1, 0))], nrec).astype(np.int64)
                         if npat else np.zeros(1, dtype=np.int64))
        rec = np.zeros(nrec, dtype=REC_DTYPE)
        rec["dst"] = -1
        rec["dst_t"] = -1
        rec["slots"] = 0xFFFFFFFF
        # ---- contributions -> (element, pair, transposed) ----
        start = plan.blk_ptr[tb]
        tot = int(t_cnt.sum())
        owner = np.repeat(np.arange(nt, dtype=np.int64), t_cnt)               # task of every contribution
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
        code = slot | (_PAIR_I[pr] << 8) | (_PAIR_J[pr] << 10) | (tr << 12) | (np.uint32(1) << 13)
        # contribution c of a task sits at half-lane hbase + c (lane = half >> 1, step = half & 1)
        chalf = hbase[owner] + within
        rec["ev"][chalf >> 1, chalf & 1] = code
        # ---- roles ----
        lane0 = hbase >> 1
        nl = np.where(hw >= 2, (np.maximum(t_cnt, 1) + 1) // 2, 1)            # lanes of the task (half tasks: 1)
        last = lane0 + nl - 1                                                  # lane that emits the block
        half_task = hw == 1
        which = np.where(half_task, hbase & 1, 0)                              # output 0 / 1 of that lane
        off_diag = (I[tb] < J[tb]) & has_partner[tb]
        dst, dst_t = dst_all[tb], np.where(off_diag, dst_all[pidx[tb]], -1)
        rec["dst"][last, which], rec["stride"][last, which] = dst, stride_all[tb]
        rec["dst_t"][last, which], rec["stride_t"][last, which] = dst_t, np.where(off_diag, stride_all[pidx[tb]], 0)
        fl = np.zeros(nrec, dtype=np.uint32)
        np.bitwise_or.at(fl, last, np.where(which == 0, FL_OUT0, FL_OUT1).astype(np.uint32))
        np.bitwise_or.at(fl, last[~half_task & (t_cnt < 4)], np.uint32(FL_PAD))
        np.bitwise_or.at(fl, last[half_task], np.uint32(FL_PAD))
        np.bitwise_or.at(fl, lane0[long_t], np.uint32(FL_ZERO_START))
        multi = np.flatnonzero(nl > 1)
        if multi.size:                                                         # chained lanes: position 1 .. nl-1
            reps = nl[multi] - 1
            cl = np.repeat(lane0[multi], reps) + (np.arange(int(reps.sum())) - np.repeat(np.cumsum(reps) - reps, reps)) + 1
            cpos = cl - np.repeat(lane0[multi], reps)
            fl[cl] |= np.uint32(FL_CHAIN) | (cpos.astype(np.uint32) << FL_POS_SHIFT)
            if np.any((cl & 15) == 0):
                raise AssertionError("a chain crosses a 16-lane row")
        # longest chain of every wave (uniform over the wave): the kernel's propagation loop count
        wave_of = np.arange(nrec) >> 6
        mc = np.ones(max(nrec >> 6, 1), dtype=np.uint32)
        np.maximum.at(mc, lane0 >> 6, nl.astype(np.uint32))
        fl |= mc[wave_of] << FL_MAXCHAIN_SHIFT
        rec["flags"] = fl
        # ---- staging slots and store order of the blocks a wave emits (direct and transposed copies run through memory
        # in address order) ----
        tt = np.flatnonzero(dst_t >= 0)
        o_task = np.concatenate([np.arange(nt), tt])
        o_addr = np.concatenate([dst, dst_t[tt]])
        o_kind = np.concatenate([np.zeros(nt, dtype=np.int64), np.ones(tt.shape[0], dtype=np.int64)])
        o_wave = last[o_task] >> 6
        # inside a wave: every emitted block gets a staging slot, numbered by (output 0 / 1, address of the direct copy);
        # the transposed copies are read back from the same slot and are ranked on their own, by address of the copy
        o_class = np.where(o_kind == 0, which[o_task], 0)
        oo = np.lexsort((o_addr, o_class, o_kind, o_wave))
        ws = o_wave[oo] * 2 + o_kind[oo]
        srank = np.arange(oo.shape[0], dtype=np.int64) - np.searchsorted(ws, ws)
        if srank.size and srank.max() > 254:
            raise ValueError("more than 255 blocks stored by one wave")
        sl = rec["slots"].copy()
        shift = (8 * (2 * which[o_task[oo]] + o_kind[oo])).astype(np.uint32)
        ln = last[o_task[oo]]
        np.bitwise_and.at(sl, ln, ~(np.uint32(255) << shift))
        np.bitwise_or.at(sl, ln, srank.astype(np.uint32) << shift)
        rec["slots"] = sl
        self.recs = rec
        self.n_patches = npat
        self.n_tasks = int(nt)
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
                "stage2_lane_fill": self.n_evals / max(int(rounds.sum()) * 512, 1),
                "stage2_wave_fill": self.n_evals / max(2 * self.recs.shape[0], 1),
                "record_bytes_per_element": self.recs.nbytes / max(self.E, 1)}

    def to_device(self, device):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        return {"pat_eptr": t(self.pat_eptr), "pat_elems": t(self.pat_elems), "pat_rptr": t(self.pat_rptr),
                "recs": t(self.recs.view(np.uint8).reshape(-1))}
