// femshell_fused_plan.cpp -- host side of fs_fused_assemble_quad: the per-mesh patch plan (include/femshell_hip.h,
// "Fused evaluate-and-assemble").  Integer bookkeeping only: which elements a patch of nodes touches, which CSR
// node-pair blocks it produces, which (element, local node pair) contributions every block is made of, in ELEMENT
// ORDER (the reference's ADD_VALUES insertion order, core/assembler.py:295-300), and on which lane each contribution
// is evaluated.  Host arrays in, host arrays out; the caller uploads them.  OpenMP over patches.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "femshell_hip.h"
#include "femshell_internal.h"

struct fs_fused_plan {
    std::vector<int64_t> pat_eptr, pat_rptr;
    std::vector<int32_t> pat_elems;
    fs_fused_rec *recs = nullptr;      // n_recs records, NOT value-initialised (gigabytes at 10 M elements: every
    int64_t n_recs = 0;                // record is written by the parallel copy below)
    int64_t n_evals = 0;
    ~fs_fused_plan() { std::free(recs); }
};

namespace {

struct Mesh {
    const int64_t *conn;
    int64_t E, N;
    const int64_t *nbr_ptr;
    const int32_t *nbr;
    std::vector<int64_t> n2e_ptr;
    std::vector<int32_t> n2e;   // elements of every node, ascending
};

struct Task {          // one CSR block (I, J) produced by a patch
    int64_t I, J;
    int64_t dst, dst_t;
    int32_t stride, stride_t;
    int n;             // contributions
    uint32_t ev[32];   // slot | ti << 8 | tj << 10 | transposed << 12
};

// node -> elements touching it, ascending element index, every element at most once per node.  All threads: counts with
// atomic increments, fill with atomic cursors (any order), then every node's short list is sorted and deduplicated.
inline void build_node_to_elements(const int64_t *conn, int64_t E, int nn, int64_t n_nodes, std::vector<int64_t> &ptr,
                                   std::vector<int32_t> &n2e) {
    std::vector<int64_t> cnt(n_nodes + 1, 0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < E * nn; i++) {
#pragma omp atomic
        cnt[conn[i] + 1]++;
    }
    for (int64_t i = 0; i < n_nodes; i++) cnt[i + 1] += cnt[i];
    std::vector<int32_t> raw((size_t)E * nn);
    std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; e++)
        for (int k = 0; k < nn; k++) {
            int64_t pos;
#pragma omp atomic capture
            pos = cur[conn[e * nn + k]]++;
            raw[pos] = (int32_t)e;
        }
    std::vector<int64_t> len(n_nodes + 1, 0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_nodes; i++) {
        int32_t *b = raw.data() + cnt[i], *e = raw.data() + cnt[i + 1];
        std::sort(b, e);
        len[i + 1] = std::unique(b, e) - b;
    }
    for (int64_t i = 0; i < n_nodes; i++) len[i + 1] += len[i];
    ptr.swap(len);
    if (ptr[n_nodes] == E * nn) { n2e.swap(raw); return; }
    n2e.resize(ptr[n_nodes]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_nodes; i++)
        std::copy(raw.begin() + cnt[i], raw.begin() + cnt[i] + (ptr[i + 1] - ptr[i]), n2e.begin() + ptr[i]);
}

// Nodes that lead a block row, sorted by (patch id, node index), and the range of every patch in that order.  Patch ids
// are any integers: dense small ones (the usual case: 0 .. n_patches - 1) go through a counting sort, O(N); anything
// else through a stable comparison sort.
inline void nodes_by_patch(int64_t n_nodes, const int64_t *nbr_ptr, const int64_t *node_patch, std::vector<int64_t> &order,
                           std::vector<int64_t> &pstart) {
    int64_t lo = INT64_MAX, hi = INT64_MIN, n_lead = 0;
    for (int64_t i = 0; i < n_nodes; i++)
        if (nbr_ptr[i + 1] > nbr_ptr[i]) { lo = std::min(lo, node_patch[i]); hi = std::max(hi, node_patch[i]); n_lead++; }
    order.clear();
    pstart.clear();
    if (n_lead == 0) { pstart.push_back(0); return; }
    if (lo >= 0 && hi < 4 * n_nodes + 1024) {
        std::vector<int64_t> cnt((size_t)hi + 2, 0);
        for (int64_t i = 0; i < n_nodes; i++)
            if (nbr_ptr[i + 1] > nbr_ptr[i]) cnt[node_patch[i] + 1]++;
        for (int64_t q = 0; q <= hi; q++) {
            if (cnt[q + 1] > 0) pstart.push_back(cnt[q]);
            cnt[q + 1] += cnt[q];
        }
        pstart.push_back(n_lead);
        order.resize(n_lead);
        for (int64_t i = 0; i < n_nodes; i++)             // ascending node index inside a patch
            if (nbr_ptr[i + 1] > nbr_ptr[i]) order[cnt[node_patch[i]]++] = i;
        return;
    }
    order.reserve(n_lead);
    for (int64_t i = 0; i < n_nodes; i++)
        if (nbr_ptr[i + 1] > nbr_ptr[i]) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return node_patch[a] < node_patch[b]; });
    for (int64_t i = 0; i < (int64_t)order.size(); i++)
        if (i == 0 || node_patch[order[i]] != node_patch[order[i - 1]]) pstart.push_back(i);
    pstart.push_back((int64_t)order.size());
}

// position of `key` in the sorted range [b, e) or -1
inline int64_t find_sorted(const int32_t *b, const int32_t *e, int32_t key) {
    const int32_t *p = std::lower_bound(b, e, key);
    return (p != e && *p == key) ? (p - b) : -1;
}

// Enumerates the tasks of one patch in (I, J) order, i.e. in ascending dst.  Returns 0, or a negative code:
// -1 a block has more than 32 contributions, -2 a contribution's element is missing from the patch list,
// -3 a block with contributions has no transposed partner (connectivity not symmetric).
template <class F>
int patch_tasks(const Mesh &m, const int64_t *nodes, int64_t n_nodes, const int32_t *elems, int ne, F &&emit) {
    Task t;
    for (int64_t q = 0; q < n_nodes; q++) {
        const int64_t I = nodes[q];
        const int64_t r0 = m.nbr_ptr[I], r1 = m.nbr_ptr[I + 1];
        const int32_t mI = (int32_t)(r1 - r0);
        const int32_t *eI = m.n2e.data() + m.n2e_ptr[I], *eIe = m.n2e.data() + m.n2e_ptr[I + 1];
        for (int64_t b = r0; b < r1; b++) {
            const int64_t J = m.nbr[b];
            const int32_t *rowJ = m.nbr + m.nbr_ptr[J], *rowJe = m.nbr + m.nbr_ptr[J + 1];
            const int64_t posI = find_sorted(rowJ, rowJe, (int32_t)I);
            const bool partner = posI >= 0;
            if (I > J && partner) continue;            // written as the transposed copy of (J, I) by J's patch
            t.I = I; t.J = J;
            t.dst = 36 * r0 + 6 * (b - r0);
            t.stride = 6 * mI;
            t.dst_t = (I < J && partner) ? 36 * m.nbr_ptr[J] + 6 * posI : -1;
            t.stride_t = 6 * (int32_t)(rowJe - rowJ);
            t.n = 0;
            // elements holding both I and J, ascending
            const int32_t *eJ = m.n2e.data() + m.n2e_ptr[J], *eJe = m.n2e.data() + m.n2e_ptr[J + 1];
            const int32_t *a = eI, *c = eJ;
            while (a != eIe && c != eJe) {
                if (*a < *c) a++;
                else if (*c < *a) c++;
                else {
                    const int32_t e = *a;
                    const int64_t *cn = m.conn + (int64_t)e * 4;
                    int li = -1, lj = -1;
                    for (int k = 0; k < 4; k++) { if (cn[k] == I && li < 0) li = k; if (cn[k] == J && lj < 0) lj = k; }
                    const int64_t slot = find_sorted(elems, elems + ne, e);
                    if (slot < 0) return -2;
                    if (t.n >= 32) return -1;
                    const int ti = li < lj ? li : lj, tj = li < lj ? lj : li;
                    t.ev[t.n++] = (uint32_t)slot | ((uint32_t)ti << 8) | ((uint32_t)tj << 10) | ((li > lj ? 1u : 0u) << 12);
                    a++; c++;
                }
            }
            if (t.n > 0 && !partner) return -3;
            emit(t);
        }
    }
    return 0;
}

typedef Task TaskRec;

// Layout of one patch in the record format of include/femshell_hip.h (two contributions per lane).  `tasks` arrive in
// (I, J) order.  Returns the number of lanes (a multiple of 64); with rec != nullptr also fills the records, which
// start at global lane index `base` (a multiple of 64: waves are numbered globally).
int64_t layout_patch(std::vector<TaskRec> &tasks, fs_fused_rec *rec, std::vector<int> &scratch) {
    const int nt = (int)tasks.size();
    // leading nodes in ascending order -> normal groups of 12 nodes; long tasks (n > 4) -> extra groups, 4 per group
    std::vector<int> &grp = scratch;
    grp.assign(nt, 0);
    int nlead = 0, nlong = 0;
    {
        int64_t prev = -1;
        for (int t = 0; t < nt; t++) {
            if (tasks[t].I != prev) { prev = tasks[t].I; nlead++; }
            grp[t] = (nlead - 1) / 12;
        }
    }
    const int gnorm = (nlead + 11) / 12;
    for (int t = 0; t < nt; t++)
        if (tasks[t].n > 4) { grp[t] = gnorm + nlong / 4; nlong++; }
    const int ngroups = gnorm + (nlong + 3) / 4;
    // order inside a group: long rows (already in (I, J) order), then 2-lane tasks, 1-lane tasks, half-lane tasks
    auto hw_of = [](int n) { return n > 4 ? 32 : n >= 3 ? 4 : n == 2 ? 2 : 1; };
    auto cls_of = [](int n) { return n > 4 ? 0 : n >= 3 ? 1 : n == 2 ? 2 : 3; };
    std::vector<int> ord(nt);
    for (int t = 0; t < nt; t++) ord[t] = t;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
        if (grp[a] != grp[b]) return grp[a] < grp[b];
        return cls_of(tasks[a].n) < cls_of(tasks[b].n);
    });
    // half-lane position of every task
    std::vector<int64_t> hbase(nt);
    int64_t lanes = 0;
    {
        int i = 0;
        for (int gidx = 0; gidx < ngroups; gidx++) {
            int64_t half = 0;
            for (; i < nt && grp[ord[i]] == gidx; i++) { hbase[ord[i]] = 2 * lanes + half; half += hw_of(tasks[ord[i]].n); }
            lanes += ((half + 1) / 2 + 63) / 64 * 64;
        }
    }
    if (!rec) return lanes;
    struct Out { int64_t addr; int lane; int kind; int which; };
    std::vector<Out> outs;
    outs.reserve(2 * nt);
    std::vector<uint32_t> mc(lanes / 64, 1u);
    for (int t = 0; t < nt; t++) {
        const TaskRec &k = tasks[t];
        const int hw = hw_of(k.n);
        const int64_t hb = hbase[t], lane0 = hb >> 1;
        for (int c = 0; c < k.n; c++) rec[(hb + c) >> 1].ev[(hb + c) & 1] = k.ev[c] | (1u << 13);
        const int nl = hw >= 2 ? ((k.n > 1 ? k.n : 1) + 1) / 2 : 1;
        const int64_t last = lane0 + nl - 1;
        const bool half_task = hw == 1;
        const int which = half_task ? (int)(hb & 1) : 0;
        rec[last].dst[which] = k.dst; rec[last].stride[which] = k.stride;
        rec[last].dst_t[which] = k.dst_t; rec[last].stride_t[which] = k.dst_t >= 0 ? k.stride_t : 0;
        rec[last].flags |= which == 0 ? FS_FL_OUT0 : FS_FL_OUT1;
        if (half_task || k.n < 4) rec[last].flags |= FS_FL_PAD;
        if (k.n > 4) rec[lane0].flags |= FS_FL_ZERO_START;
        for (int c = 1; c < nl; c++) rec[lane0 + c].flags |= FS_FL_CHAIN | ((uint32_t)c << 16);
        if ((uint32_t)nl > mc[lane0 >> 6]) mc[lane0 >> 6] = (uint32_t)nl;
        outs.push_back({k.dst, (int)last, 0, which});
        if (k.dst_t >= 0) outs.push_back({k.dst_t, (int)last, 1, which});
    }
    for (int64_t l = 0; l < lanes; l++) rec[l].flags |= mc[l >> 6] << 8;
    // staging slots: the blocks a wave emits get a slot each, numbered by (output 0 / 1, address of the direct copy);
    // the transposed copies are read back from the same slot and get a rank of their own, by address of the copy
    std::sort(outs.begin(), outs.end(), [](const Out &a, const Out &b) {
        const int wa = a.lane >> 6, wb = b.lane >> 6;
        if (wa != wb) return wa < wb;
        if (a.kind != b.kind) return a.kind < b.kind;
        const int ca = a.kind == 0 ? a.which : 0, cb = b.kind == 0 ? b.which : 0;
        if (ca != cb) return ca < cb;
        return a.addr < b.addr;
    });
    int rank = 0;
    for (size_t i = 0; i < outs.size(); i++) {
        rank = (i > 0 && (outs[i].lane >> 6) == (outs[i - 1].lane >> 6) && outs[i].kind == outs[i - 1].kind) ? rank + 1 : 0;
        if (rank > 254) return -1;
        const uint32_t shift = 8u * (2u * (uint32_t)outs[i].which + (uint32_t)outs[i].kind);
        uint32_t &sl = rec[outs[i].lane].slots;
        sl = (sl & ~(255u << shift)) | ((uint32_t)rank << shift);
    }
    return lanes;
}

}  // namespace

extern "C" {

int fs_fused_plan_build(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                        const int64_t *node_patch, int32_t max_patch_elems, fs_fused_plan **plan, int64_t *n_patches,
                        int64_t *n_elem_slots, int64_t *n_recs) {
    if (!conn || !nbr_ptr || !nbr || !node_patch || !plan || !n_patches || !n_elem_slots || !n_recs)
        return fs_fail_(FS_EINVAL, "fs_fused_plan_build: null argument", 0);
    if (E < 0 || n_nodes <= 0 || E > INT32_MAX) return fs_fail_(FS_EINVAL, "fs_fused_plan_build: bad sizes", 0);
    if (max_patch_elems < 16 || max_patch_elems > FS_FUSED_MAX_PATCH_ELEMS)
        return fs_fail_(FS_EINVAL, "fs_fused_plan_build: max_patch_elems must lie in [16, FS_FUSED_MAX_PATCH_ELEMS]", 0);
    {
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (int64_t i = 0; i < E * 4; i++) bad |= (conn[i] < 0 || conn[i] >= n_nodes) ? 1 : 0;
        if (bad) return fs_fail_(FS_EINVAL, "fs_fused_plan_build: connectivity out of range", 0);
    }
    fs_fused_plan *P = new (std::nothrow) fs_fused_plan;
    if (!P) return fs_fail_(FS_EINVAL, "fs_fused_plan_build: out of memory", 0);
    try {
        Mesh m{conn, E, n_nodes, nbr_ptr, nbr, {}, {}};
        // node -> elements (ascending; a node listed twice by one element counts once)
        build_node_to_elements(conn, E, 4, n_nodes, m.n2e_ptr, m.n2e);
        // nodes that lead a block row, sorted by (patch, node)
        std::vector<int64_t> order, pstart;        // pstart: ranges of `order`, one per patch
        nodes_by_patch(n_nodes, nbr_ptr, node_patch, order, pstart);
        const int64_t np = (int64_t)pstart.size() - 1;
        std::vector<int64_t> ne(np, 0), nl(np, 0), nev(np, 0);
        std::vector<std::vector<int32_t>> els(np);
        std::vector<std::vector<fs_fused_rec>> prec(np);
        fs_fused_rec blank;
        std::memset(&blank, 0, sizeof blank);
        blank.dst[0] = blank.dst[1] = blank.dst_t[0] = blank.dst_t[1] = -1;
        blank.slots = 0xFFFFFFFFu;
        int err = 0;
        // one pass over the patches: element list, tasks, lane layout and records into per-patch buffers ...
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t p = 0; p < np; p++) {
            const int64_t *nodes = order.data() + pstart[p];
            const int64_t nn = pstart[p + 1] - pstart[p];
            std::vector<int32_t> &el = els[p];
            for (int64_t q = 0; q < nn; q++)
                el.insert(el.end(), m.n2e.begin() + m.n2e_ptr[nodes[q]], m.n2e.begin() + m.n2e_ptr[nodes[q] + 1]);
            std::sort(el.begin(), el.end());
            el.erase(std::unique(el.begin(), el.end()), el.end());
            ne[p] = (int64_t)el.size();
            if ((int64_t)el.size() > max_patch_elems) {
#pragma omp atomic write
                err = -4;
                continue;
            }
            std::vector<TaskRec> tasks;
            std::vector<int> scratch;
            int64_t evs = 0;
            const int rc = patch_tasks(m, nodes, nn, el.data(), (int)el.size(), [&](const Task &t) { tasks.push_back(t); evs += t.n; });
            if (rc) {
#pragma omp atomic write
                err = rc;
                continue;
            }
            nl[p] = layout_patch(tasks, nullptr, scratch);
            nev[p] = evs;
            prec[p].assign((size_t)nl[p], blank);
            if (layout_patch(tasks, prec[p].data(), scratch) < 0) {
#pragma omp atomic write
                err = -5;
            }
        }
        if (err) {
            delete P;
            const char *msg = err == -4 ? "fs_fused_plan_build: a patch touches more than max_patch_elems elements"
                            : err == -1 ? "fs_fused_plan_build: a node pair is shared by more than 32 elements"
                            : err == -3 ? "fs_fused_plan_build: block with contributions but no transposed partner (pattern not symmetric)"
                            : err == -5 ? "fs_fused_plan_build: more than 255 blocks stored by one wave"
                                        : "fs_fused_plan_build: inconsistent connectivity";
            return fs_fail_(FS_EINVAL, msg, 0);
        }
        P->pat_eptr.assign(np + 1, 0);
        P->pat_rptr.assign(np + 1, 0);
        for (int64_t p = 0; p < np; p++) {
            P->pat_eptr[p + 1] = P->pat_eptr[p] + ne[p];
            P->pat_rptr[p + 1] = P->pat_rptr[p] + nl[p];
            P->n_evals += nev[p];
        }
        P->pat_elems.resize(P->pat_eptr[np]);
        P->n_recs = P->pat_rptr[np];
        P->recs = static_cast<fs_fused_rec *>(std::malloc((size_t)(P->n_recs > 0 ? P->n_recs : 1) * sizeof(fs_fused_rec)));
        if (!P->recs) throw std::bad_alloc();
        // ... then into place
#pragma omp parallel for schedule(static)
        for (int64_t p = 0; p < np; p++) {
            std::copy(els[p].begin(), els[p].end(), P->pat_elems.begin() + P->pat_eptr[p]);
            if (!prec[p].empty()) std::memcpy(P->recs + P->pat_rptr[p], prec[p].data(), prec[p].size() * sizeof(fs_fused_rec));
            std::vector<fs_fused_rec>().swap(prec[p]);
        }
        *plan = P;
        *n_patches = np;
        *n_elem_slots = P->pat_eptr[np];
        *n_recs = P->pat_rptr[np];
        return FS_OK;
    } catch (const std::bad_alloc &) {
        delete P;
        return fs_fail_(FS_EINVAL, "fs_fused_plan_build: out of memory", 0);
    }
}

int fs_fused_plan_export(const fs_fused_plan *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_rptr,
                         fs_fused_rec *recs, int64_t *n_evals) {
    if (!plan || !pat_eptr || !pat_elems || !pat_rptr || !recs)
        return fs_fail_(FS_EINVAL, "fs_fused_plan_export: null argument", 0);
    std::copy(plan->pat_eptr.begin(), plan->pat_eptr.end(), pat_eptr);
    std::copy(plan->pat_rptr.begin(), plan->pat_rptr.end(), pat_rptr);
    std::copy(plan->pat_elems.begin(), plan->pat_elems.end(), pat_elems);
    {   // gigabytes: copied by all threads
        const int64_t n = plan->n_recs, chunk = 1 << 16;
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < (n + chunk - 1) / chunk; c++) {
            const int64_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
            std::memcpy(recs + lo, plan->recs + lo, (size_t)(hi - lo) * sizeof(fs_fused_rec));
        }
    }
    if (n_evals) *n_evals = plan->n_evals;
    return FS_OK;
}

void fs_fused_plan_free(fs_fused_plan *plan) { delete plan; }

}  // extern "C"


// ------------------------------------------------------------------------------------------------------------------
// Plan of fs_fused_assemble_hex8 (include/femshell_hip.h, "Fused evaluate-and-assemble, HEXA8"): patches of nodes; a
// patch forms the COMPLETE 3-row CSR slabs of its nodes (all 27 node-pair blocks of an interior node, not only J >= I),
// so that its output leaves as long contiguous runs.  A lane works through up to 8 STEPS, one contribution (element, local
// node pair) each, accumulating a block's contributions in ELEMENT ORDER (0 + x0 + x1 + ..., the reference's ADD_VALUES
// insertion order, core/assembler.py:295-300) and flushing the finished block; the blocks of a node are packed into
// lanes first-fit by decreasing contribution count (interior node of a structured mesh: 8 | 4+4 x 3 | 2+2+2+2 x 3 |
// 8 x 1, each padded to 2 steps = 9 lanes of 8 steps).
// ------------------------------------------------------------------------------------------------------------------
struct fs_fused_hex_plan {
    std::vector<int64_t> pat_eptr, pat_nptr, pat_rptr, pat_lptr;
    std::vector<int32_t> pat_elems;
    std::vector<fs_hex_node> nodes;
    std::vector<fs_hex_run> runs;
    std::vector<uint32_t> steps;      // [lanes][8]
    int64_t n_evals = 0;
};

namespace {

// one CSR block (I, J) of a node: contributions slot | la << 7 | lb << 10 with la / lb the local index of the SMALLER /
// LARGER of the two global nodes in that element; tr = the block is the transpose of that ordered pair's (I > J)
struct HexBlock { int rank; int n; uint32_t tr; uint32_t ev[8]; };

// LDS slots of a patch's elements that make phase 2's gathers conflict-free on structured patches.  When the (ascending)
// element indices form a box ix + s2 iy + s3 iz (the numbering of a structured block), element (ix, iy, iz) gets a slot
// congruent to ix + 4 iy + g(iz) modulo 16 -- the 16-byte bank slot of its gradients in LDS.  The 16 lanes of a hardware
// ds_read_b128 group then hold the k-th lanes of a 4 x 4 square of nodes of one layer (hex_patch_layout orders them so),
// whose elements form a 4 x 4 square as well: 16 consecutive residues, no two on one bank.  The layer shifts g are
// chosen greedily so that the slots stay dense (5 x 5 x 4 elements: shifts 0, 5, 10, 15 fill slots 0..99 exactly); a
// smaller box leaves a few slots unused (they repeat the first element).  Returns false (slots = positions in the
// ascending list) for anything else.
bool lattice_slots(const int32_t *el, int ne, int max_slots, std::vector<int> &slot_of, int &n_slots) {
    if (ne < 8) return false;
    int nx = 1;
    while (nx < ne && el[nx] == el[0] + nx) nx++;
    if (nx > 8 || ne % nx) return false;
    const int rows = ne / nx;
    const int64_t s2 = rows > 1 ? (int64_t)el[nx] - el[0] : 0;
    int ny = 1;
    while (ny < rows && (int64_t)el[ny * nx] == el[0] + ny * s2) ny++;
    if (rows % ny) return false;
    const int nz = rows / ny;
    const int64_t s3 = nz > 1 ? (int64_t)el[ny * nx] - el[0] : 0;
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++)
                if ((int64_t)el[(iz * ny + iy) * nx + ix] != el[0] + ix + iy * s2 + iz * s3) return false;
    int m[16] = {0};
    for (int iy = 0; iy < ny; iy++)
        for (int ix = 0; ix < nx; ix++) m[(ix + 4 * iy) & 15]++;
    auto need = [](const int *load) {
        int P = 0;
        for (int r = 0; r < 16; r++)
            if (load[r] > 0 && (load[r] - 1) * 16 + r + 1 > P) P = (load[r] - 1) * 16 + r + 1;
        return P;
    };
    int load[16] = {0}, shift[16];
    if (nz > 16) return false;
    for (int iz = 0; iz < nz; iz++) {
        int best_g = 0, best_need = 1 << 30, best_max = 1 << 30;
        for (int g = 0; g < 16; g++) {
            int l[16], mx = 0;
            for (int r = 0; r < 16; r++) { l[r] = load[r] + m[(r - g) & 15]; if (l[r] > mx) mx = l[r]; }
            const int nd = need(l);
            if (nd < best_need || (nd == best_need && mx < best_max)) { best_need = nd; best_max = mx; best_g = g; }
        }
        shift[iz] = best_g;
        for (int r = 0; r < 16; r++) load[r] += m[(r - best_g) & 15];
    }
    n_slots = need(load);
    if (n_slots > max_slots) return false;
    int next_free[16];
    for (int r = 0; r < 16; r++) next_free[r] = r;
    slot_of.assign(ne, 0);
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                const int r = (ix + 4 * iy + shift[iz]) & 15;
                slot_of[(iz * ny + iy) * nx + ix] = next_free[r];
                next_free[r] += 16;
            }
    return true;
}

// returns 0 or a negative code (-1: a node pair shared by more than 8 elements, -2: element missing from the patch,
// -6: limits of the step record exceeded)
int hex_patch_layout(const Mesh &m, const int64_t *conn, const int64_t *nodes, int nn, const int32_t *elems, int ne,
                     const int *slot_of, std::vector<fs_hex_node> *out_nodes, std::vector<fs_hex_run> *out_runs, std::vector<uint32_t> *out_steps,
                     int64_t *n_lanes, int64_t *n_evals, int64_t *stage_len) {
    if (nn > FS_HEX_MAX_PATCH_NODES || ne > 128) return -6;
    int64_t lanes = 0, evals = 0, sbase = 0;
    std::vector<HexBlock> blocks;
    std::vector<int> ord;
    for (int q = 0; q < nn; q++) {
        const int64_t I = nodes[q];
        const int64_t r0 = m.nbr_ptr[I], r1 = m.nbr_ptr[I + 1];
        const int mI = (int)(r1 - r0);
        if (mI > 128) return -6;
        const int32_t *eI = m.n2e.data() + m.n2e_ptr[I], *eIe = m.n2e.data() + m.n2e_ptr[I + 1];
        blocks.clear();
        for (int64_t b = r0; b < r1; b++) {
            const int64_t J = m.nbr[b];
            HexBlock hb;
            hb.rank = (int)(b - r0);
            hb.n = 0;
            hb.tr = I > J ? (1u << 28) : 0u;
            const int32_t *eJ = m.n2e.data() + m.n2e_ptr[J], *eJe = m.n2e.data() + m.n2e_ptr[J + 1];
            const int32_t *a = eI, *c = eJ;
            while (a != eIe && c != eJe) {
                if (*a < *c) a++;
                else if (*c < *a) c++;
                else {
                    const int32_t e = *a;
                    const int64_t *cn = conn + (int64_t)e * 8;
                    int li = -1, lj = -1;
                    for (int k = 0; k < 8; k++) { if (cn[k] == I && li < 0) li = k; if (cn[k] == J && lj < 0) lj = k; }
                    const int64_t at = find_sorted(elems, elems + ne, e);
                    const int64_t slot = at < 0 ? -1 : (slot_of ? slot_of[at] : at);
                    if (slot < 0) return -2;
                    if (hb.n >= 8) return -1;
                    hb.ev[hb.n++] = (uint32_t)slot | ((uint32_t)(I > J ? lj : li) << 7) | ((uint32_t)(I > J ? li : lj) << 10) | hb.tr;
                    a++; c++;
                }
            }
            evals += hb.n;
            blocks.push_back(hb);
        }
        // first fit, decreasing: a block takes max(n, 1) steps rounded up to an EVEN number (a pattern-only block still
        // flushes its zeros), so that blocks start at even steps and flush at odd ones: the kernel keeps a finished block
        // in one of FOUR register sets (index = step / 2) until the phase ends
        ord.resize(blocks.size());
        for (size_t i = 0; i < ord.size(); i++) ord[i] = (int)i;
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return blocks[x].n > blocks[y].n; });
        std::vector<int> fill;                                 // steps used per lane of this node
        const int64_t lane0 = lanes;
        for (int bi : ord) {
            const HexBlock &hb = blocks[bi];
            const int need = ((hb.n > 0 ? hb.n : 1) + 1) & ~1;
            size_t l = 0;
            while (l < fill.size() && fill[l] + need > 8) l++;
            if (l == fill.size()) {
                fill.push_back(0);
                if (out_steps) out_steps->resize((size_t)(lane0 + (int64_t)fill.size()) * 8, 0u);
            }
            if (out_steps) {
                uint32_t *st = out_steps->data() + (size_t)(lane0 + (int64_t)l) * 8 + fill[l];
                for (int c = 0; c < need; c++) {
                    uint32_t r = c < hb.n ? (hb.ev[c] | (1u << 13)) : 0u;
                    if (c == need - 1) r |= (1u << 14) | ((uint32_t)q << 15) | ((uint32_t)hb.rank << 21) | hb.tr;
                    st[c] = r;
                }
            }
            fill[l] += need;
        }
        lanes += (int64_t)fill.size();
        if (out_nodes) out_nodes->push_back(fs_hex_node{(int32_t)sbase, (int32_t)mI});
        if (out_runs) {
            const int64_t gdst = 9 * r0;
            if (!out_runs->empty() && out_runs->back().gdst + out_runs->back().len == gdst &&
                (int64_t)out_runs->back().sbase + out_runs->back().len == sbase)
                out_runs->back().len += 9 * mI;
            else
                out_runs->push_back(fs_hex_run{gdst, (int32_t)sbase, (int32_t)(9 * mI)});
        }
        sbase += 9 * (int64_t)mI;
    }
    *n_lanes = (lanes + 63) / 64 * 64;
    if (out_steps) {
        out_steps->resize((size_t)*n_lanes * 8, 0u);
        // Lane order.  Phase 2 of the kernel gathers its operands from LDS with ds_read_b128, which the hardware serves in
        // groups of 16 lanes: {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of every 32.  Lanes of ONE node name unrelated
        // (element, local node) pairs -- a random gather, 3 LDS cycles per group instead of 1 --, whereas the k-th lanes
        // of CONSECUTIVE nodes of a structured patch name element slots that differ like the nodes do: so the lanes are
        // ordered k-major (k = position of the lane among its node's lanes), nodes ascending inside a k, and every run of
        // 16 goes to one hardware group.  Which lane forms a block is invisible in the result (a flush names its node
        // and neighbour rank).
        std::vector<std::pair<int, int>> key;                  // (k, first lane of the node) of every lane
        key.reserve((size_t)lanes);
        {
            const uint32_t *st = out_steps->data();
            int prev_q = -1, k = 0;
            for (int64_t l = 0; l < lanes; l++) {
                int q = -1;                                    // node of the lane: the one its flushes name
                for (int c = 0; c < 8 && q < 0; c++)
                    if ((st[l * 8 + c] >> 14) & 1u) q = (int)((st[l * 8 + c] >> 15) & 63u);
                k = (q == prev_q) ? k + 1 : 0;
                prev_q = q;
                key.push_back({k, q});
            }
        }
        std::vector<int64_t> ord((size_t)lanes);
        for (int64_t l = 0; l < lanes; l++) ord[l] = l;
        std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
        static const int group_lane[32] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27,
                                           4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
        std::vector<uint32_t> re((size_t)*n_lanes * 8, 0u);
        for (int64_t i = 0; i < lanes; i++) {
            const int64_t dst = (i / 32) * 32 + group_lane[i % 32];
            std::copy(out_steps->begin() + ord[i] * 8, out_steps->begin() + ord[i] * 8 + 8, re.begin() + dst * 8);
        }
        out_steps->swap(re);
    }
    *n_evals = evals;
    *stage_len = sbase;
    return 0;
}

}  // namespace

extern "C" {

int fs_fused_hex_plan_build(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                            const int64_t *node_patch, fs_fused_hex_plan **plan, int64_t *n_patches, int64_t *n_elem_slots,
                            int64_t *n_pat_nodes, int64_t *n_runs, int64_t *n_lanes) {
    if (!conn || !nbr_ptr || !nbr || !node_patch || !plan || !n_patches || !n_elem_slots || !n_pat_nodes || !n_runs || !n_lanes)
        return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_build: null argument", 0);
    if (E < 0 || n_nodes <= 0 || E > INT32_MAX) return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_build: bad sizes", 0);
    for (int64_t i = 0; i < E * 8; i++)
        if (conn[i] < 0 || conn[i] >= n_nodes) return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_build: connectivity out of range", 0);
    fs_fused_hex_plan *P = new (std::nothrow) fs_fused_hex_plan;
    if (!P) return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_build: out of memory", 0);
    try {
        Mesh m{conn, E, n_nodes, nbr_ptr, nbr, {}, {}};
        build_node_to_elements(conn, E, 8, n_nodes, m.n2e_ptr, m.n2e);
        if (m.n2e_ptr[n_nodes] != E * 8) {
            delete P;
            return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_build: an element lists a node twice", 0);
        }
        std::vector<int64_t> order, pstart;
        nodes_by_patch(n_nodes, nbr_ptr, node_patch, order, pstart);
        const int64_t np = (int64_t)pstart.size() - 1;
        std::vector<int64_t> ne(np, 0), nl(np, 0), nev(np, 0), nr(np, 0);
        std::vector<std::vector<int32_t>> els(np);
        std::vector<std::vector<fs_hex_run>> runs(np);
        std::vector<std::vector<fs_hex_node>> pnodes(np);
        std::vector<std::vector<uint32_t>> steps(np);
        int err = 0;
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t p = 0; p < np; p++) {
            const int64_t *nodes = order.data() + pstart[p];
            const int nn = (int)(pstart[p + 1] - pstart[p]);
            std::vector<int32_t> &el = els[p];
            for (int q = 0; q < nn; q++)
                el.insert(el.end(), m.n2e.begin() + m.n2e_ptr[nodes[q]], m.n2e.begin() + m.n2e_ptr[nodes[q] + 1]);
            std::sort(el.begin(), el.end());
            el.erase(std::unique(el.begin(), el.end()), el.end());
            ne[p] = (int64_t)el.size();
            int rc = 0;
            int64_t stage = 0;
            if ((int64_t)el.size() > FS_HEX_MAX_PATCH_ELEMS) rc = -4;
            else {
                std::vector<int> slot_of;
                int n_slots = (int)el.size();
                const bool lat = lattice_slots(el.data(), (int)el.size(), FS_HEX_MAX_PATCH_ELEMS, slot_of, n_slots);
                rc = hex_patch_layout(m, conn, nodes, nn, el.data(), (int)el.size(), lat ? slot_of.data() : nullptr, &pnodes[p],
                                      &runs[p], &steps[p], &nl[p], &nev[p], &stage);
                if (lat) {                                   // pat_elems in slot order; unused slots repeat the first element
                    std::vector<int32_t> by_slot((size_t)n_slots, el[0]);
                    for (size_t i = 0; i < el.size(); i++) by_slot[slot_of[i]] = el[i];
                    el.swap(by_slot);
                    ne[p] = (int64_t)el.size();
                }
            }
            if (!rc && (nl[p] > FS_HEX_MAX_PATCH_LANES || stage > FS_HEX_STAGE_DOUBLES)) rc = -7;
            if (rc) {
#pragma omp atomic write
                err = rc;
            }
        }
        if (err) {
            delete P;
            const char *msg = err == -4 ? "fs_fused_hex_plan_build: a patch touches more than FS_HEX_MAX_PATCH_ELEMS elements"
                            : err == -1 ? "fs_fused_hex_plan_build: a node pair is shared by more than 8 elements"
                            : err == -7 ? "fs_fused_hex_plan_build: a patch needs more than FS_HEX_MAX_PATCH_LANES lanes or staging doubles"
                            : err == -6 ? "fs_fused_hex_plan_build: more than 64 nodes in a patch or more than 128 neighbours of a node"
                                        : "fs_fused_hex_plan_build: inconsistent connectivity";
            return fs_fail_(FS_EINVAL, msg, 0);
        }
        P->pat_eptr.assign(np + 1, 0); P->pat_nptr.assign(np + 1, 0); P->pat_rptr.assign(np + 1, 0); P->pat_lptr.assign(np + 1, 0);
        for (int64_t p = 0; p < np; p++) {
            P->pat_eptr[p + 1] = P->pat_eptr[p] + ne[p];
            P->pat_nptr[p + 1] = P->pat_nptr[p] + (int64_t)pnodes[p].size();
            P->pat_rptr[p + 1] = P->pat_rptr[p] + (int64_t)runs[p].size();
            P->pat_lptr[p + 1] = P->pat_lptr[p] + nl[p];
            P->n_evals += nev[p];
        }
        P->pat_elems.resize(P->pat_eptr[np]);
        P->nodes.resize(P->pat_nptr[np]);
        P->runs.resize(P->pat_rptr[np]);
        P->steps.resize((size_t)P->pat_lptr[np] * 8);
#pragma omp parallel for schedule(static)
        for (int64_t p = 0; p < np; p++) {
            std::copy(els[p].begin(), els[p].end(), P->pat_elems.begin() + P->pat_eptr[p]);
            std::copy(pnodes[p].begin(), pnodes[p].end(), P->nodes.begin() + P->pat_nptr[p]);
            std::copy(runs[p].begin(), runs[p].end(), P->runs.begin() + P->pat_rptr[p]);
            std::copy(steps[p].begin(), steps[p].end(), P->steps.begin() + (size_t)P->pat_lptr[p] * 8);
        }
        *plan = P;
        *n_patches = np;
        *n_elem_slots = P->pat_eptr[np];
        *n_pat_nodes = P->pat_nptr[np];
        *n_runs = P->pat_rptr[np];
        *n_lanes = P->pat_lptr[np];
        return FS_OK;
    } catch (const std::bad_alloc &) {
        delete P;
        return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_build: out of memory", 0);
    }
}

int fs_fused_hex_plan_export(const fs_fused_hex_plan *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_nptr,
                             fs_hex_node *nodes, int64_t *pat_rptr, fs_hex_run *runs, int64_t *pat_lptr, uint32_t *steps,
                             int64_t *n_evals) {
    if (!plan || !pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_rptr || !runs || !pat_lptr || !steps)
        return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_export: null argument", 0);
    std::copy(plan->pat_eptr.begin(), plan->pat_eptr.end(), pat_eptr);
    std::copy(plan->pat_nptr.begin(), plan->pat_nptr.end(), pat_nptr);
    std::copy(plan->pat_rptr.begin(), plan->pat_rptr.end(), pat_rptr);
    std::copy(plan->pat_lptr.begin(), plan->pat_lptr.end(), pat_lptr);
    std::copy(plan->pat_elems.begin(), plan->pat_elems.end(), pat_elems);
    std::copy(plan->nodes.begin(), plan->nodes.end(), nodes);
    std::copy(plan->runs.begin(), plan->runs.end(), runs);
    std::copy(plan->steps.begin(), plan->steps.end(), steps);
    if (n_evals) *n_evals = plan->n_evals;
    return FS_OK;
}

void fs_fused_hex_plan_free(fs_fused_hex_plan *plan) { delete plan; }

}  // extern "C"


// ------------------------------------------------------------------------------------------------------------------
// Plan of fs_fused_assemble_tri (include/femshell_hip.h, "Fused evaluate-and-assemble, MITC3+"): the step layout of the
// HEXA8 plan with 6 x 6 blocks that leave from registers -- no staging image, no copy-out runs; a node record names where
// the node's CSR slab begins.  Interior node of a structured triangulation: 6 | 2 x 6 contributions = 3 lanes of 6 steps.
// ------------------------------------------------------------------------------------------------------------------
struct fs_fused_tri_plan {
    std::vector<int64_t> pat_eptr, pat_nptr, pat_lptr;
    std::vector<int32_t> pat_elems;
    std::vector<fs_tri_node> nodes;
    std::vector<uint32_t> steps;      // [lanes][8]
    int64_t n_evals = 0;
};

namespace {

struct TriBlock { int q; int rank; int n; uint32_t ev[8]; };    // contributions slot | li << 8 | lj << 10 (li of I, lj of J)

// returns 0 or a negative code (-1: a node pair shared by more than 8 elements, -2: element missing from the patch,
// -6: limits of the step record exceeded)
int tri_patch_layout(const Mesh &m, const int64_t *conn, const int64_t *nodes, int nn, const int32_t *elems, int ne,
                     std::vector<fs_tri_node> *out_nodes, std::vector<uint32_t> *out_steps, int64_t *n_lanes, int64_t *n_evals) {
    if (nn > FS_TRI_MAX_PATCH_NODES || ne > FS_TRI_MAX_PATCH_ELEMS) return -6;
    int64_t evals = 0;
    std::vector<TriBlock> blocks;
    for (int q = 0; q < nn; q++) {
        const int64_t I = nodes[q];
        const int64_t r0 = m.nbr_ptr[I], r1 = m.nbr_ptr[I + 1];
        const int mI = (int)(r1 - r0);
        if (mI > 128) return -6;
        const int32_t *eI = m.n2e.data() + m.n2e_ptr[I], *eIe = m.n2e.data() + m.n2e_ptr[I + 1];
        for (int64_t b = r0; b < r1; b++) {
            const int64_t J = m.nbr[b];
            TriBlock tb;
            tb.q = q;
            tb.rank = (int)(b - r0);
            tb.n = 0;
            const int32_t *eJ = m.n2e.data() + m.n2e_ptr[J], *eJe = m.n2e.data() + m.n2e_ptr[J + 1];
            const int32_t *a = eI, *c = eJ;
            while (a != eIe && c != eJe) {
                if (*a < *c) a++;
                else if (*c < *a) c++;
                else {
                    const int32_t e = *a;
                    const int64_t *cn = conn + (int64_t)e * 3;
                    int li = -1, lj = -1;
                    for (int k = 0; k < 3; k++) { if (cn[k] == I && li < 0) li = k; if (cn[k] == J && lj < 0) lj = k; }
                    const int64_t slot = find_sorted(elems, elems + ne, e);
                    if (slot < 0) return -2;
                    if (tb.n >= 8) return -1;
                    tb.ev[tb.n++] = (uint32_t)slot | ((uint32_t)li << 8) | ((uint32_t)lj << 10);
                    a++; c++;
                }
            }
            evals += tb.n;
            blocks.push_back(tb);
        }
        if (out_nodes) out_nodes->push_back(fs_tri_node{36 * r0, (int32_t)mI, 0});
    }
    // The blocks of the whole patch, first fit by decreasing contribution count (stable: node order survives inside a
    // class, so neighbouring lanes store neighbouring blocks).  A block takes max(n, 1) steps (a pattern-only block
    // flushes its zeros); lanes are filled to SIX steps -- 6 | 2 + 2 + 2 on a structured triangulation: every lane full,
    // all four waves of the workgroup busy -- and only a block of 7 or 8 contributions uses the last two.
    std::vector<int> ord(blocks.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = (int)i;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return blocks[x].n > blocks[y].n; });
    std::vector<int> fill;
    size_t first_open = 0;                               // lanes before it are full
    for (int bi : ord) {
        const TriBlock &tb = blocks[bi];
        const int need = tb.n > 0 ? tb.n : 1;
        const int cap = need > 6 ? 8 : 6;
        size_t l = first_open;
        while (l < fill.size() && fill[l] + need > cap) l++;
        if (l == fill.size()) {
            fill.push_back(0);
            if (out_steps) out_steps->resize(fill.size() * 8, 0u);
        }
        if (out_steps) {
            uint32_t *st = out_steps->data() + l * 8 + fill[l];
            for (int c = 0; c < need; c++) {
                uint32_t r = c < tb.n ? (tb.ev[c] | (1u << 13)) : 0u;
                if (c == need - 1) r |= (1u << 14) | ((uint32_t)tb.q << 15) | ((uint32_t)tb.rank << 22);
                st[c] = r;
            }
        }
        fill[l] += need;
        while (first_open < fill.size() && fill[first_open] >= 6) first_open++;
    }
    *n_lanes = ((int64_t)fill.size() + 63) / 64 * 64;
    if (out_steps) out_steps->resize((size_t)*n_lanes * 8, 0u);
    *n_evals = evals;
    return 0;
}

}  // namespace

extern "C" {

int fs_fused_tri_plan_build(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                            const int64_t *node_patch, fs_fused_tri_plan **plan, int64_t *n_patches, int64_t *n_elem_slots,
                            int64_t *n_pat_nodes, int64_t *n_lanes) {
    if (!conn || !nbr_ptr || !nbr || !node_patch || !plan || !n_patches || !n_elem_slots || !n_pat_nodes || !n_lanes)
        return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_build: null argument", 0);
    if (E < 0 || n_nodes <= 0 || E > INT32_MAX) return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_build: bad sizes", 0);
    for (int64_t i = 0; i < E * 3; i++)
        if (conn[i] < 0 || conn[i] >= n_nodes) return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_build: connectivity out of range", 0);
    fs_fused_tri_plan *P = new (std::nothrow) fs_fused_tri_plan;
    if (!P) return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_build: out of memory", 0);
    try {
        Mesh m{conn, E, n_nodes, nbr_ptr, nbr, {}, {}};
        build_node_to_elements(conn, E, 3, n_nodes, m.n2e_ptr, m.n2e);
        if (m.n2e_ptr[n_nodes] != E * 3) {
            delete P;
            return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_build: an element lists a node twice", 0);
        }
        std::vector<int64_t> order, pstart;
        nodes_by_patch(n_nodes, nbr_ptr, node_patch, order, pstart);
        const int64_t np = (int64_t)pstart.size() - 1;
        std::vector<int64_t> ne(np, 0), nl(np, 0), nev(np, 0);
        std::vector<std::vector<int32_t>> els(np);
        std::vector<std::vector<fs_tri_node>> pnodes(np);
        std::vector<std::vector<uint32_t>> steps(np);
        int err = 0;
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t p = 0; p < np; p++) {
            const int64_t *nodes = order.data() + pstart[p];
            const int nn = (int)(pstart[p + 1] - pstart[p]);
            std::vector<int32_t> &el = els[p];
            for (int q = 0; q < nn; q++)
                el.insert(el.end(), m.n2e.begin() + m.n2e_ptr[nodes[q]], m.n2e.begin() + m.n2e_ptr[nodes[q] + 1]);
            std::sort(el.begin(), el.end());
            el.erase(std::unique(el.begin(), el.end()), el.end());
            ne[p] = (int64_t)el.size();
            int rc = 0;
            if ((int64_t)el.size() > FS_TRI_MAX_PATCH_ELEMS) rc = -4;
            else rc = tri_patch_layout(m, conn, nodes, nn, el.data(), (int)el.size(), &pnodes[p], &steps[p], &nl[p], &nev[p]);
            if (!rc && nl[p] > FS_TRI_MAX_PATCH_LANES) rc = -7;
            if (rc) {
#pragma omp atomic write
                err = rc;
            }
        }
        if (err) {
            delete P;
            const char *msg = err == -4 ? "fs_fused_tri_plan_build: a patch touches more than FS_TRI_MAX_PATCH_ELEMS elements"
                            : err == -1 ? "fs_fused_tri_plan_build: a node pair is shared by more than 8 elements"
                            : err == -7 ? "fs_fused_tri_plan_build: a patch needs more than FS_TRI_MAX_PATCH_LANES lanes"
                            : err == -6 ? "fs_fused_tri_plan_build: too many nodes in a patch or more than 128 neighbours of a node"
                                        : "fs_fused_tri_plan_build: inconsistent connectivity";
            return fs_fail_(FS_EINVAL, msg, 0);
        }
        P->pat_eptr.assign(np + 1, 0); P->pat_nptr.assign(np + 1, 0); P->pat_lptr.assign(np + 1, 0);
        for (int64_t p = 0; p < np; p++) {
            P->pat_eptr[p + 1] = P->pat_eptr[p] + ne[p];
            P->pat_nptr[p + 1] = P->pat_nptr[p] + (int64_t)pnodes[p].size();
            P->pat_lptr[p + 1] = P->pat_lptr[p] + nl[p];
            P->n_evals += nev[p];
        }
        P->pat_elems.resize(P->pat_eptr[np]);
        P->nodes.resize(P->pat_nptr[np]);
        P->steps.resize((size_t)P->pat_lptr[np] * 8);
#pragma omp parallel for schedule(static)
        for (int64_t p = 0; p < np; p++) {
            std::copy(els[p].begin(), els[p].end(), P->pat_elems.begin() + P->pat_eptr[p]);
            std::copy(pnodes[p].begin(), pnodes[p].end(), P->nodes.begin() + P->pat_nptr[p]);
            std::copy(steps[p].begin(), steps[p].end(), P->steps.begin() + (size_t)P->pat_lptr[p] * 8);
        }
        *plan = P;
        *n_patches = np;
        *n_elem_slots = P->pat_eptr[np];
        *n_pat_nodes = P->pat_nptr[np];
        *n_lanes = P->pat_lptr[np];
        return FS_OK;
    } catch (const std::bad_alloc &) {
        delete P;
        return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_build: out of memory", 0);
    }
}

int fs_fused_tri_plan_export(const fs_fused_tri_plan *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_nptr,
                             fs_tri_node *nodes, int64_t *pat_lptr, uint32_t *steps, int64_t *n_evals) {
    if (!plan || !pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_lptr || !steps)
        return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_export: null argument", 0);
    std::copy(plan->pat_eptr.begin(), plan->pat_eptr.end(), pat_eptr);
    std::copy(plan->pat_nptr.begin(), plan->pat_nptr.end(), pat_nptr);
    std::copy(plan->pat_lptr.begin(), plan->pat_lptr.end(), pat_lptr);
    std::copy(plan->pat_elems.begin(), plan->pat_elems.end(), pat_elems);
    std::copy(plan->nodes.begin(), plan->nodes.end(), nodes);
    std::copy(plan->steps.begin(), plan->steps.end(), steps);
    if (n_evals) *n_evals = plan->n_evals;
    return FS_OK;
}

void fs_fused_tri_plan_free(fs_fused_tri_plan *plan) { delete plan; }

}  // extern "C"
