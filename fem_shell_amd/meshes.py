"""Array-native meshes for the hot path.

`ArrayMesh` is the batch-friendly mesh container (coordinates [N,3], connectivity [E,nn]); it also
exposes the duck-typed protocol MeshAssembler touches in the reference (`elements`, `node_count`,
`node_id_to_index`, `element_sets`; core/assembler.py:72,107,113-119,220) through lazy views, so
no per-element Python object exists unless a caller asks for one.

Generators restate the structured mesh builders of the reference's own benchmarks
(tests/benchmarks/test_ko2017_performance.py:857-893 cylindrical patch, :261-308 flat patch) as
array code, plus the synthetic C5 plate of SURVEY.md 8(d).
"""
from types import SimpleNamespace

import numpy as np


class _ElemView:
    __slots__ = ("_m", "id")

    def __init__(self, mesh, idx):
        self._m, self.id = mesh, idx

    @property
    def node_ids(self):
        return tuple(int(i) for i in self._m.conn[self.id] if i >= 0)

    @property
    def node_count(self):
        return int((self._m.conn[self.id] >= 0).sum())

    @property
    def node_coords(self):
        row = self._m.conn[self.id]
        return self._m.coords[row[row >= 0]]

    thickness = None


class _ElemList:
    def __init__(self, mesh, index=None):
        self._m, self._index = mesh, index

    def __len__(self):
        return self._m.conn.shape[0] if self._index is None else len(self._index)

    def __getitem__(self, i):
        return _ElemView(self._m, int(i if self._index is None else self._index[i]))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __bool__(self):
        return len(self) > 0


class ArrayMesh:
    """coords [N,3] float64, conn [E,nn] int (node indices == node ids), optional element sets
    given as {name: int array of element indices} or a per-element set-id array + names.
    Mixed node counts (e.g. triangles + quads): pad the short rows of `conn` with -1."""

    def __init__(self, coords, conn, set_id=None, set_names=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int64)
        self.set_id = None if set_id is None else np.ascontiguousarray(set_id, dtype=np.int32)
        self.set_names = list(set_names) if set_names is not None else None
        if self.set_id is not None and self.set_names is None:
            self.set_names = [f"set_{i}" for i in range(int(self.set_id.max()) + 1)]

    @property
    def node_count(self):
        return self.coords.shape[0]

    @property
    def element_count(self):
        return self.conn.shape[0]

    @property
    def elements(self):
        return _ElemList(self)

    @property
    def node_id_to_index(self):
        return _Identity(self.node_count)

    @property
    def element_sets(self):
        if self.set_id is None:
            return {}
        return {name: SimpleNamespace(elements=_ElemList(self, np.flatnonzero(self.set_id == i)))
                for i, name in enumerate(self.set_names)}


class _Identity:
    def __init__(self, n):
        self.n = n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise KeyError(i)
        return i

    def __len__(self):
        return self.n


def hash_uniform(ids, salt):
    """splitmix64 of (ids, salt) -> float64 in [0,1); counter-based so it is order independent."""
    with np.errstate(over="ignore"):
        z = ids.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(salt + 1)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def structured_conn(nx, ny, triangular=False):
    """Quads (n00,n10,n11,n01), row-major in j then i; triangles split (n00,n10,n11),(n00,n11,n01)
    exactly like the reference benchmarks (test_ko2017_performance.py:296-306, 882-892)."""
    ids = np.arange((nx + 1) * (ny + 1), dtype=np.int64).reshape(ny + 1, nx + 1)
    n00, n10, n11, n01 = ids[:-1, :-1], ids[:-1, 1:], ids[1:, 1:], ids[1:, :-1]
    if not triangular:
        return np.stack([n00, n10, n11, n01], -1).reshape(-1, 4)
    t = np.stack([np.stack([n00, n10, n11], -1), np.stack([n00, n11, n01], -1)], 2)
    return t.reshape(-1, 3)


def mixed_conn(nx, ny):
    """Quads and triangle pairs alternating cell by cell (checkerboard), triangles padded with -1:
    the smallest mesh that exercises a mixed MITC3+/MITC4+ assembly."""
    q = structured_conn(nx, ny)
    cell = np.arange(nx * ny)
    tri_cell = ((cell // nx + cell % nx) % 2).astype(bool)
    rows = []
    for c in range(nx * ny):
        n00, n10, n11, n01 = q[c]
        if tri_cell[c]:
            rows.append((n00, n10, n11, -1))
            rows.append((n00, n11, n01, -1))
        else:
            rows.append((n00, n10, n11, n01))
    return np.array(rows, dtype=np.int64)


def cylindrical_patch(radius, length, angle_deg, nx, ny, triangular=False):
    """test_ko2017_performance.py:857-893 (_build_cylindrical_patch)."""
    zs = np.linspace(0.0, length, ny + 1)
    th = np.linspace(0.0, np.radians(angle_deg), nx + 1)
    x = radius * np.cos(th)
    y = radius * np.sin(th)
    coords = np.empty((ny + 1, nx + 1, 3))
    coords[:, :, 0] = x[None, :]
    coords[:, :, 1] = y[None, :]
    coords[:, :, 2] = zs[:, None]
    return ArrayMesh(coords.reshape(-1, 3), structured_conn(nx, ny, triangular))


def flat_patch(x0, x1, y0, y1, nx, ny, triangular=False, surface_z=None):
    """test_ko2017_performance.py:261-308 (_build_structured_mesh_xy)."""
    xs = np.linspace(x0, x1, nx + 1)
    ys = np.linspace(y0, y1, ny + 1)
    X, Y = np.meshgrid(xs, ys)
    Z = np.zeros_like(X) if surface_z is None else surface_z(X, Y)
    return ArrayMesh(np.stack([X, Y, Z], -1).reshape(-1, 3), structured_conn(nx, ny, triangular))


def jittered_plate(nx, ny, cell=1e-3, seed=11, n_sets=16, x_offset_cells=0, band=None):
    """Synthetic C5 laminate plate: nx x ny MITC4+ cells of size `cell`, nodes
    jittered +-5 % in-plane and +-1 % out-of-plane so every element is distinct and warped;
    `n_sets` property sets striped by x-band. `x_offset_cells` shifts the strip (rank shards).
    band=None: ceil(nx / n_sets) columns per band, counted from the piece's own first column; band=k: bands of k
    columns of the GLOBAL plate cycling through the sets, so that an x-strip is exactly its part of the whole plate."""
    ix = np.arange(nx + 1, dtype=np.int64) + x_offset_cells
    iy = np.arange(ny + 1, dtype=np.int64)
    IX, IY = np.meshgrid(ix, iy)
    X, Y = IX * cell, IY * cell
    coords = np.stack([X, Y, np.zeros_like(X)], -1).reshape(-1, 3)
    # jitter is a pure function of the GLOBAL grid position, so shards generated on different
    # ranks agree bit for bit on their shared interface nodes
    gid = (IY * (1 << 32) + IX).reshape(-1)
    coords[:, 0] += (hash_uniform(gid, seed) - 0.5) * 0.1 * cell
    coords[:, 1] += (hash_uniform(gid, seed + 1) - 0.5) * 0.1 * cell
    coords[:, 2] += (hash_uniform(gid, seed + 2) - 0.5) * 0.02 * cell
    conn = structured_conn(nx, ny)
    if band is None:
        set_id = np.tile((np.arange(nx) // max(1, -(-nx // n_sets))).astype(np.int32) % n_sets, ny)
    else:
        set_id = np.tile(((np.arange(nx) + x_offset_cells) // int(band)).astype(np.int32) % n_sets, ny)
    return ArrayMesh(coords, conn, set_id, [f"band_{i}" for i in range(n_sets)])


def twisted_beam(length, width, twist_deg, nx, ny, triangular=False):
    """MacNeal-Harder twisted beam (test_ko2017_performance.py:1195-1229): a flat strip along x whose
    cross-section rotates linearly by `twist_deg` from root to tip."""
    mesh = flat_patch(0.0, length, -width / 2.0, width / 2.0, nx, ny, triangular)
    phi = np.radians(twist_deg) * (mesh.coords[:, 0] / length)
    y = mesh.coords[:, 1].copy()
    mesh.coords[:, 1] = y * np.cos(phi)
    mesh.coords[:, 2] = y * np.sin(phi)
    return mesh


def sphere_patch(radius, theta_min_deg, theta_max_deg, phi_max_deg, n, triangular=False):
    """test_ko2017_performance.py:1558-1604 (_build_sphere_patch): structured in (theta, phi); the
    hemisphere with an 18 degree cut-out is theta in [18, 90], phi in [0, 90]."""
    th = np.linspace(np.radians(theta_min_deg), np.radians(theta_max_deg), n + 1)
    ph = np.linspace(0.0, np.radians(phi_max_deg), n + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")           # node (i, j) -> row j (theta), column i (phi)
    coords = np.stack([radius * np.sin(TH) * np.cos(PH), radius * np.sin(TH) * np.sin(PH), radius * np.cos(TH)], -1)
    return ArrayMesh(coords.reshape(-1, 3), structured_conn(n, n, triangular))


def tapered_tube(n_around, n_span, length=60.0, r_root=2.5, r_tip=0.4, n_sets_around=24, n_sets_span=8):
    """Blade-like closed shell for config C3: a tapered, slightly twisted and flattened tube of
    n_around x n_span quads (closed in the circumferential direction), split into n_sets_around x n_sets_span
    element sets.  The real blade geometry cannot be regenerated here (no gmsh / turbine YAML)."""
    s = np.linspace(0.0, 1.0, n_span + 1)
    a = np.arange(n_around) * (2.0 * np.pi / n_around)
    S, A = np.meshgrid(s, a, indexing="ij")
    r = r_root + (r_tip - r_root) * S
    tw = np.radians(15.0) * S
    xc, yc = r * np.cos(A), 0.45 * r * np.sin(A)                       # flattened section
    coords = np.stack([xc * np.cos(tw) - yc * np.sin(tw), xc * np.sin(tw) + yc * np.cos(tw), length * S], -1).reshape(-1, 3)
    ids = np.arange((n_span + 1) * n_around, dtype=np.int64).reshape(n_span + 1, n_around)
    nxt = np.roll(ids, -1, axis=1)
    conn = np.stack([ids[:-1], nxt[:-1], nxt[1:], ids[1:]], -1).reshape(-1, 4)
    ia = np.tile(np.arange(n_around) * n_sets_around // n_around, n_span)
    js = np.repeat(np.arange(n_span) * n_sets_span // n_span, n_around)
    set_id = (js * n_sets_around + ia).astype(np.int32)
    return ArrayMesh(coords, conn, set_id, [f"set_{k}" for k in range(n_sets_around * n_sets_span)])
