"""MeshAssembler: the reference's global-assembler interface (core/assembler.py:12-516) on top of
the batched HIP kernels.

Same constructor `(mesh, model)`, same `model["elements"]` dictionary (element_family, material,
thickness, laminate, properties, extra element kwargs), same attributes (`dofs_per_node`,
`spatial_dim`, `dofs_count`, `_row_nnz`, `_dofs_array`, `_ke_array`, `_me_array`, `_element_map`,
`_node_dofs_map`, `_is_mixed_mesh`) and the same assemble_* methods.  Differences, by design:
  * matrices come back as `CsrMatrix` (device CSR: indptr/indices/data torch tensors) with
    `.to_scipy()` and `.to_petsc()`; where petsc4py exists `.to_petsc()` yields the AIJ Mat the
    unchanged solver layer expects (core/assembler.py:243-303);
  * element matrices live on the GPU in FS_LAYOUT_SYMBLK; `_ke_array` / `_me_array` /
    `_dofs_array` / `_element_map` are materialised lazily;
  * degenerate elements raise ValueError (stricter than MITC4's silent NaN).
"""
import warnings
from typing import Dict, Optional

import ctypes

import numpy as np
import torch

from . import _lib, batch
from .elements import ElementFactory, ElementFamily
from .properties import (CompositeShellProperty, ShellProperty, build_prop_table, iso_row,
                         laminate_row, solid_row, PROP_DTYPE)
from .symbolic import BlockPlan, DeviceBlockPlan, can_tile, vector_plan


class CsrMatrix:
    """Device CSR (rows = DOFs, sorted columns, explicit structural zeros kept, like PETSc AIJ
    assembled without MAT_IGNORE_ZERO_ENTRIES)."""

    def __init__(self, indptr, indices, data, shape):
        self.indptr, self.indices, self.data, self.shape = indptr, indices, data, tuple(shape)

    @property
    def nnz(self):
        return int(self.data.numel())

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.data.cpu().numpy(), self.indices.cpu().numpy(),
                              self.indptr.cpu().numpy()), shape=self.shape)

    def to_petsc(self, comm=None):
        try:
            from petsc4py import PETSc
        except ImportError as exc:
            raise ImportError("petsc4py is not installed: use CsrMatrix.to_scipy() or the device "
                              "tensors directly") from exc
        ip = self.indptr.cpu().numpy().astype(PETSc.IntType)
        ix = self.indices.cpu().numpy().astype(PETSc.IntType)
        return PETSc.Mat().createAIJWithArrays(self.shape, (ip, ix, self.data.cpu().numpy()), comm=comm)

    def lump_rows(self):
        """Row-sum lumping (solvers/fsi/linear_dynamic.py:227-234) as a device vector."""
        from .bc import lump_mass_matrix
        return lump_mass_matrix(self)


class _LazyElementMap:
    """dict-like view element id -> element object (built on demand; core/assembler.py:190)."""

    def __init__(self, asm):
        self._a, self._cache = asm, {}

    def __len__(self):
        return len(self._a._elem_ids)

    def __iter__(self):
        return iter(self._a._elem_ids.tolist())

    def keys(self):
        return self._a._elem_ids.tolist()

    def items(self):
        return ((k, self[k]) for k in self)

    def values(self):
        return (self[k] for k in self)

    def __contains__(self, k):
        return k in self._a._elem_pos

    def __getitem__(self, eid):
        if eid not in self._cache:
            self._cache[eid] = self._a._make_element(self._a._elem_pos[eid])
        return self._cache[eid]


class MeshAssembler:
    reuse_linear_blocks = True   # K_T / f_int re-assembly reads the cached linear blocks (tiled quad meshes)
    use_fused = True             # uniform MITC4+ meshes: assemble_stiffness_matrix(recompute=True) is ONE fused
                                 # evaluate-and-assemble launch (fs_fused_assemble_quad): the two-kernel result to
                                 # rounding (4e-16 of the largest entry), exactly symmetric, bitwise repeatable
    fused_max_patch_elems = 64   # elements per patch of the fused plan (<= 80)
    newton_on_assembled = True   # uniform MITC4+ meshes: K_T(u) = K_csr + element-order sums of the geometric corrections
                                 # (fs_fused_assemble_quad_tangent), f_int(u) = K_csr u + nonlinear membrane forces
                                 # (fs_csr_blk_matvec + fs_shell_quad_fint_nl): no element blocks in the Newton loop
    _FAMILY_PROPERTIES = {ElementFamily.SHELL: (6, 3), ElementFamily.PLANE: (2, 2),
                          ElementFamily.SOLID: (3, 3)}

    def __init__(self, mesh, model: Dict, comm=None, device=None, plan=None, cache_blocks=True):
        """cache_blocks=False (uniform MITC4+ meshes): the constructor does not evaluate and keep the element stiffness
        blocks (2 880 B per element) -- the geometry check of the reference's constructor loop runs through the
        status words of the fused kernel's mass pass (frame + Jacobians only), K and M come from the fused kernel, and
        `_ke_array` / the Newton path evaluate the blocks on first use."""
        self._given_plan = plan   # precomputed BlockPlan (distributed runs add ghost columns)
        self._cache_blocks = bool(cache_blocks)
        self.mesh = mesh
        self.model = model["elements"]
        self.comm = comm
        self.device = torch.device(device) if device is not None else batch._dev()
        self.dofs_per_node = 0
        self.spatial_dim = 0
        self.dofs_count = 0
        self._row_nnz: Optional[np.ndarray] = None
        self._is_mixed_mesh = False
        self._K_blk = self._M_blk = None
        self._K_lin_csr = None    # values of the assembled linear K the Newton path works on (see _linear_csr)
        self._K_lin_packed = None # (those values, their packed upper block triangle): see _linear_packed
        self._fplan = self._fplan_d = None
        self._fplan_m = self._fplan_m_d = None
        self._fplan_error = None
        self._precompute_elements()
        self._compute_sparsity_pattern()

    # ------------------------------------------------------------------ input extraction
    def _extract(self):
        mesh = self.mesh
        if hasattr(mesh, "coords") and hasattr(mesh, "conn"):      # ArrayMesh fast path
            coords, conn = mesh.coords, mesh.conn
            elem_ids = np.arange(conn.shape[0], dtype=np.int64)
            thickness = None
            set_lookup = None
            self._node_ids = None          # node id == node index
            if getattr(mesh, "set_id", None) is not None:
                set_lookup = (mesh.set_id, mesh.set_names)
            return coords, conn, elem_ids, thickness, set_lookup
        # duck-typed reference mesh (core/assembler.py:72,107,121-187)
        n2i = mesh.node_id_to_index
        E, nnodes = len(mesh.elements), max(e.node_count for e in mesh.elements)
        conn = np.full((E, nnodes), -1, dtype=np.int64)     # short rows padded with -1
        coords = np.full((mesh.node_count, 3), np.nan)
        elem_ids = np.empty(E, dtype=np.int64)
        thickness = np.full(E, np.nan)
        self._node_ids = np.full(mesh.node_count, -1, dtype=np.int64)      # node index -> node id
        for i, el in enumerate(mesh.elements):
            idx = [n2i[n] for n in el.node_ids]
            conn[i, :len(idx)] = idx
            self._node_ids[idx] = list(el.node_ids)
            coords[idx] = np.asarray(el.node_coords, dtype=float)
            elem_ids[i] = el.id
            if getattr(el, "thickness", None) is not None:
                thickness[i] = el.thickness
        return coords, conn, elem_ids, (None if np.isnan(thickness).all() else thickness), None

    def _property_ids(self, E, elem_ids, thickness, set_lookup):
        """Rows of the device property table + per-element row index (core/assembler.py:111-162)."""
        m = self.model
        extra = {k: v for k, v in m.items()
                 if k not in ("properties", "material", "thickness", "laminate", "element_family")}
        nonlinear = bool(extra.get("nonlinear", False))
        kappa = extra.get("shear_correction_factor")
        rows, pid = [], np.full(E, -1, dtype=np.int32)
        self._prop_objects = []

        def add(row, obj):
            rows.append(row)
            self._prop_objects.append(obj)
            return len(rows) - 1

        props_map = m.get("properties")
        if props_map is not None:
            if set_lookup is not None:
                set_id, names = set_lookup
                for sid, name in enumerate(names):
                    if name in props_map:
                        p = props_map[name]
                        r = add(laminate_row(p.laminate, nonlinear) if isinstance(p, CompositeShellProperty)
                                else iso_row(p.material, p.thickness, kappa, nonlinear), p)
                        pid[set_id == sid] = r
            else:
                pos = {int(e): i for i, e in enumerate(elem_ids)}
                for name, p in props_map.items():
                    if name in self.mesh.element_sets:
                        r = add(laminate_row(p.laminate, nonlinear) if isinstance(p, CompositeShellProperty)
                                else iso_row(p.material, p.thickness, kappa, nonlinear), p)
                        for el in self.mesh.element_sets[name].elements:
                            if int(el.id) in pos:          # skipped (unsupported) elements have no row
                                pid[pos[int(el.id)]] = r
        rest = pid < 0
        if rest.any():
            if m.get("laminate") is not None:
                pid[rest] = add(laminate_row(m["laminate"], nonlinear), CompositeShellProperty(m["laminate"]))
            else:
                if "material" not in m:
                    raise TypeError("model['elements'] needs 'material'+'thickness', 'laminate' or 'properties'")
                if thickness is not None and np.isfinite(thickness[rest]).any():
                    warnings.warn("Per-element thickness via MeshElement.thickness is deprecated. Use a "
                                  "'properties' dict mapping element-set names to ShellProperty / "
                                  "CompositeShellProperty instead.", DeprecationWarning, stacklevel=1)
                    th = np.where(np.isfinite(thickness), thickness, m.get("thickness", np.nan))
                    for t in np.unique(th[rest]):
                        pid[rest & (th == t)] = add(iso_row(m["material"], float(t), kappa, nonlinear),
                                                    ShellProperty(m["material"], float(t)))
                else:
                    pid[rest] = add(iso_row(m["material"], m["thickness"], kappa, nonlinear),
                                    ShellProperty(m["material"], m["thickness"]))
        table = np.zeros(len(rows), dtype=PROP_DTYPE)
        for i, r in enumerate(rows):
            table[i] = r
        return table, pid

    # ------------------------------------------------------------------ precompute
    def _precompute_elements(self):
        fam = self.model.get("element_family")
        if fam not in (ElementFamily.SHELL, ElementFamily.SOLID):
            raise NotImplementedError("fem_shell_amd.MeshAssembler: ElementFamily.SHELL and .SOLID are on "
                                      "the GPU path (PLANE and family-less mixed models are not)")
        self._family = fam
        self.dofs_per_node, self.spatial_dim = self._FAMILY_PROPERTIES[fam]
        coords, conn, elem_ids, thickness, set_lookup = self._extract()
        # Elements whose node count the family's factory does not know are skipped, like the reference
        # (ElementFactory returns False on the KeyError and the loop `continue`s: elements.py:258-259,
        # core/assembler.py:163-164).
        self._mesh_pos = None              # position in mesh.elements of every kept element (None: all kept)
        counts = (conn >= 0).sum(axis=1)
        known = (3, 4) if fam == ElementFamily.SHELL else (4, 5, 6, 8, 10, 13, 15, 20)
        keep = np.isin(counts, known)
        if not keep.all():
            self._mesh_pos = np.flatnonzero(keep)
            conn, elem_ids = conn[keep], elem_ids[keep]
            width = int((conn >= 0).sum(axis=1).max()) if conn.shape[0] else 0
            conn = np.ascontiguousarray(conn[:, :width])
            thickness = None if thickness is None else thickness[keep]
            if set_lookup is not None:
                set_lookup = (set_lookup[0][keep], set_lookup[1])
        E = conn.shape[0]
        self._coords, self._conn, self._elem_ids = coords, conn, elem_ids
        self._elem_pos = {int(e): i for i, e in enumerate(elem_ids)} if E < 2_000_000 else _RangePos(E)
        self.dofs_count = int(self.mesh.node_count) * self.dofs_per_node
        dev = self.device
        if fam == ElementFamily.SHELL:
            self._prop_table, self._prop_id = self._property_ids(E, elem_ids, thickness, set_lookup)
            self._props_d = batch.to_device_props(self._prop_table)
            allowed = (3, 4)
        else:
            # elements/SOLID.py:69-104: one material (+ optional 3x3 orientation) for the whole model
            if "material" not in self.model:
                raise TypeError("model['elements'] needs 'material' for ElementFamily.SOLID")
            self._reduced = bool(self.model.get("reduced_integration", False))
            self._prop_table = solid_row(self.model["material"], self.model.get("orientation"))[None]
            self._prop_id = np.zeros(E, dtype=np.int32)
            self._prop_objects = [self.model["material"]]
            self._props_d = batch.to_device_solid_props(self._prop_table)
            allowed = (4, 5, 6, 8, 10, 13, 15, 20)
        # symbolic phase on the device (fs_csr_symbolic_*); a precomputed host BlockPlan is accepted too
        # (uniform quad meshes: element matrices in the tiled layout the stiffness kernel writes fastest)
        self._plan = self._given_plan or DeviceBlockPlan(conn, int(self.mesh.node_count), self.dofs_per_node, dev,
                                                         tiled=(fam == ElementFamily.SHELL and can_tile(conn, self.dofs_per_node)))
        p = self._plan
        self._layout = _lib.FS_LAYOUT_SYMBLK_T16 if getattr(p, "tiled", False) else _lib.FS_LAYOUT_SYMBLK
        coords_d = torch.from_numpy(coords).to(dev)
        # element groups (uniform node count each); their block buffers lie back to back
        self._groups = []
        for nn, pos, Eg, boff in p.groups:
            if nn not in allowed:
                raise NotImplementedError(f"{fam.name} elements with {nn} nodes are not on the GPU path "
                                          f"(supported node counts: {allowed})")
            cg = conn if pos is None else conn[pos, :nn]
            cg_d = torch.from_numpy(np.ascontiguousarray(cg[:, :nn])).to(dev)
            pid = self._prop_id if pos is None else self._prop_id[pos]
            self._groups.append(dict(nn=nn, pos=pos, E=Eg, boff=boff, npair=nn * (nn + 1) // 2,
                                     conn_d=cg_d, xyz_d=coords_d[cg_d],        # [E_g,nn,3] gather (plumbing)
                                     pid_d=torch.from_numpy(np.ascontiguousarray(pid)).to(dev)))
        self._is_mixed_mesh = len(self._groups) > 1
        g0 = self._groups[0]
        self._conn_d, self._xyz_d, self._pid_d = g0["conn_d"], g0["xyz_d"], g0["pid_d"]
        self._plan_d = p.to_device(dev)
        self._desc_d = self._plan_d["desc"]
        # K of every element now (this is also the geometry check of the reference's constructor loop); M on first use
        # (assemble_mass_matrix / _me_array): the reference evaluates both up front, here that would only cost memory
        if self._cache_blocks or not (self.can_fuse and fam == ElementFamily.SHELL and g0["nn"] == 4):
            self.refresh_element_matrices(want_K=True, want_M=False)
        else:
            self._status = [self._geometry_status()]
            self.check_status()

    def _geometry_status(self):
        """Status words of every quad (degenerate frame, non-positive Jacobian, bad property id) without element
        matrices: fs_shell_quad_geom_status, the geometry stage of the stiffness kernel alone."""
        g = self._groups[0]
        st = torch.zeros(g["E"], dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().fs_shell_quad_geom_status(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), len(self._prop_table),
                                                         g["E"], _lib.ptr(st), _lib.stream_ptr()),
                   "fs_shell_quad_geom_status")
        return st

    def _k_blocks(self):
        """Element stiffness blocks, evaluated on first use when the constructor skipped them (cache_blocks=False)."""
        if self._K_blk is None:
            self.refresh_element_matrices(want_K=True, want_M=False)
        return self._K_blk

    def _group_view(self, buf, g):
        dd = self.dofs_per_node ** 2
        if self._layout == _lib.FS_LAYOUT_SYMBLK_T16:      # one quad group: the whole tiled buffer
            return buf
        return buf[g["boff"]:g["boff"] + g["E"] * g["npair"]].view(g["E"], g["npair"], dd)

    def refresh_element_matrices(self, want_K=True, want_M=True, check=True):
        """(Re)run the element kernels: K and M of every element, like the element loop of
        core/assembler.py:121-193 (which always evaluates both).  `check=False` skips the
        host-synchronising status test (call check_status() later)."""
        dd = self.dofs_per_node ** 2
        nb = self._plan.n_src_blocks
        if want_K and self._K_blk is None:
            self._K_blk = torch.empty((nb, dd), dtype=torch.float64, device=self.device)
        if want_M and self._M_blk is None:
            self._M_blk = torch.empty((nb, dd), dtype=torch.float64, device=self.device)
        self._status = []
        for g in self._groups:
            kw = dict(want_K=want_K, want_M=want_M, status=True, validate=False,
                      K_out=self._group_view(self._K_blk, g) if want_K else None,
                      M_out=self._group_view(self._M_blk, g) if want_M else None)
            if self._family == ElementFamily.SHELL:
                out = batch.shell_batch(g["nn"], g["xyz_d"], g["pid_d"], self._props_d, len(self._prop_table),
                                        layout=self._layout, **kw)
            else:
                out = batch.solid_batch(g["nn"], g["xyz_d"], g["pid_d"], self._props_d, len(self._prop_table),
                                        reduced=self._reduced, **kw)
            self._status.append(out["status"])
        if check:
            self.check_status()

    def check_status(self):
        for g, st in zip(self._groups, self._status):
            bad = torch.nonzero(st)
            if bad.numel():
                e = int(bad[0])
                pos = e if g["pos"] is None else int(g["pos"][e])
                raise ValueError(f"element {int(self._elem_ids[pos])}: degenerate geometry "
                                 f"(status bits {int(st[e])})")

    def _compute_sparsity_pattern(self):
        self._row_nnz = self._plan.row_nnz()
        lib, p, d = _lib.load(), self._plan, self._plan_d
        self._indptr = torch.empty(self.dofs_count + 1, dtype=torch.int64, device=self.device)
        self._indices = torch.empty(p.nnz, dtype=torch.int32, device=self.device)
        _lib.check(lib.fs_csr_pattern_blk(_lib.ptr(d["nbr_ptr"]), _lib.ptr(d["blk_row"]), _lib.ptr(d["nbr"]),
                                          p.n_nodes, p.nblk, p.d, _lib.ptr(self._indptr),
                                          _lib.ptr(self._indices), _lib.stream_ptr()), "fs_csr_pattern_blk")

    # ------------------------------------------------------------------ lazy reference views
    def _dofs_of(self, conn_rows):
        d = self.dofs_per_node
        return (conn_rows[:, :, None] * d + np.arange(d, dtype=np.int64)).reshape(conn_rows.shape[0], -1)

    @property
    def _dofs_array(self):
        """[E, nn*d] int64 for uniform meshes, None for mixed ones (core/assembler.py:203-218)."""
        return None if self._is_mixed_mesh else self._dofs_of(self._conn)

    def _dense(self, buf, uniform):
        if self._is_mixed_mesh == uniform:
            return None
        d = self.dofs_per_node
        per_group = [batch.expand_blocks(self._group_view(buf, g), g["nn"], d, self._layout, g["E"]).cpu().numpy()
                     for g in self._groups]
        if uniform:
            return per_group[0]
        out = [None] * self._conn.shape[0]
        for g, arr in zip(self._groups, per_group):
            for k, pos in enumerate(g["pos"]):
                out[int(pos)] = arr[k]
        return out

    _ke_array = property(lambda self: self._dense(self._k_blocks(), True))
    def _mass_blocks(self):
        if self._M_blk is None:
            self.refresh_element_matrices(want_K=False, want_M=True)
        return self._M_blk

    _me_array = property(lambda self: self._dense(self._mass_blocks(), True))
    _ke_list = property(lambda self: self._dense(self._k_blocks(), False))
    _me_list = property(lambda self: self._dense(self._mass_blocks(), False))

    @property
    def _dofs_list(self):
        if not self._is_mixed_mesh:
            return None
        d = self.dofs_per_node
        return [(row[row >= 0][:, None] * d + np.arange(d, dtype=np.int64)).reshape(-1) for row in self._conn]

    @property
    def _element_map(self):
        return _LazyElementMap(self)

    @property
    def _node_dofs_map(self):
        """node ID -> its DOFs `node_index * dofs_per_node + k` (core/assembler.py:166-182), for the nodes some
        element touches."""
        d = self.dofs_per_node
        used = np.unique(self._conn[self._conn >= 0])
        ids = used if self._node_ids is None else self._node_ids[used]
        return {int(i): tuple(range(int(n) * d, int(n) * d + d)) for i, n in zip(ids, used)}

    def _make_element(self, pos):
        view = self.mesh.elements[pos if self._mesh_pos is None else int(self._mesh_pos[pos])]
        prop = self._prop_objects[int(self._prop_id[pos])]
        extra = {k: v for k, v in self.model.items()
                 if k not in ("properties", "material", "thickness", "laminate", "element_family")}
        if self._family == ElementFamily.SOLID:
            return ElementFactory.get_element(ElementFamily.SOLID, view, material=self.model["material"], **extra)
        return ElementFactory.get_element(ElementFamily.SHELL, view, shell_property=prop, **extra)

    # ------------------------------------------------------------------ assembly
    def _assemble(self, blk):
        lib, p, d = _lib.load(), self._plan, self._plan_d
        data = torch.empty(p.nnz, dtype=torch.float64, device=self.device)
        _lib.check(lib.fs_csr_numeric_desc(_lib.ptr(blk), _lib.ptr(self._desc_d), _lib.ptr(d["contrib"]),
                                           p.nblk, p.d, _lib.ptr(data), _lib.stream_ptr()),
                   "fs_csr_numeric_desc")
        return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))

    # ------------------------------------------------------------------ fused evaluate-and-assemble
    @property
    def can_fuse(self):
        """The fused kernels cover uniform MITC4+ meshes (one quad group, 6 dofs per node), uniform MITC3+ meshes and
        uniform HEXA8 meshes with the full 2 x 2 x 2 rule (3 dofs per node) whose patch plan can be built (see
        fused_ready)."""
        if self._fplan_error is not None or len(self._groups) != 1 or self._groups[0]["pos"] is not None:
            return False
        if self._family == ElementFamily.SHELL:
            return self._groups[0]["nn"] in (3, 4)                     # fs_fused_assemble_tri / fs_fused_assemble_quad
        return self._groups[0]["nn"] == 8 and not self._reduced      # HEXA8, full rule: fs_fused_assemble_hex8

    def fused_plan(self):
        """Patch plan of fs_fused_assemble_quad (fem_shell_amd/fused.py), built on first use."""
        if self._fplan_d is None:
            from .fused import FusedHexPlan, FusedQuadPlan, FusedQuadPlanDevice, partition_nodes, partition_nodes_device
            # the host plan builders read the node graph only (not the contribution lists: gigabytes at 10 M elements)
            from types import SimpleNamespace
            p = self._plan
            on_device = not isinstance(p.nbr_ptr, np.ndarray)
            host_view = lambda: p if not on_device else SimpleNamespace(
                d=p.d, n_nodes=p.n_nodes, nbr_ptr=p.nbr_ptr.cpu().numpy(), nbr=p.nbr.cpu().numpy())
            if self._family == ElementFamily.SOLID:
                from .fused import partition_nodes_hex_device
                node_patch = None
                if on_device:        # structured numbering: the node -> patch map on the device (None: NumPy partition)
                    node_patch = partition_nodes_hex_device(int(self.mesh.node_count), self._groups[0]["conn_d"])
                self._fplan = FusedHexPlan(host_view(), self._conn, self._coords, node_patch=node_patch)
                self._fused_flags = 0
            elif self._groups[0]["nn"] == 3:
                from .fused import FusedTriPlan
                self._fplan = FusedTriPlan(host_view(), self._conn, self._coords)
                self._fused_flags = 0
            else:
                self._fplan = self._build_quad_plan(None)
                self._fused_flags = _lib.FS_FUSED_HAS_COUPLED if (self._prop_table["flags"] & 1).any() else 0
            self._fplan_d = self._fplan.to_device(self.device)
        return self._fplan

    def _build_quad_plan(self, tile):
        """FusedQuadPlan(Device) with the patch ids in `tile` order (fused.PATCH_TILE when None)."""
        from types import SimpleNamespace
        from .fused import FusedQuadPlan, FusedQuadPlanDevice, partition_nodes, partition_nodes_device
        p, g = self._plan, self._groups[0]
        on_device = not isinstance(p.nbr_ptr, np.ndarray)
        # structured numbering: the node -> patch map is integer arithmetic on the device (None otherwise: the
        # NumPy partition with its bisection pass takes over)
        node_patch = partition_nodes_device(int(self.mesh.node_count), g["conn_d"], self.fused_max_patch_elems,
                                            as_host=not on_device, tile=tile)
        fplan = None
        if on_device:
            # the whole plan on the device (fs_fused_plan_device_begin / _finish); a mesh beyond its LDS capacities
            # goes through the host builder, every other failure is the mesh's (fused_ready() handles it)
            if node_patch is None:
                node_patch = torch.from_numpy(partition_nodes(self._coords, self._conn, self.fused_max_patch_elems, tile=tile))
            try:
                fplan = FusedQuadPlanDevice(p, g["conn_d"], node_patch, self.fused_max_patch_elems)
            except _lib.FemShellHipError as exc:
                if "capacity" not in str(exc):
                    raise
                node_patch = node_patch.cpu().numpy()
        if fplan is None:
            host = p if not on_device else SimpleNamespace(d=p.d, n_nodes=p.n_nodes, nbr_ptr=p.nbr_ptr.cpu().numpy(),
                                                           nbr=p.nbr.cpu().numpy())
            fplan = FusedQuadPlan(host, self._conn, self._coords, max_patch_elems=self.fused_max_patch_elems,
                                  node_patch=node_patch)
        return fplan

    def _mass_plan(self):
        """(plan, device arrays) the fused MASS operator of a quad mesh runs on: the same patches numbered row by row
        (fused.PATCH_TILE explains why), built on first use; the stiffness plan itself when that is row by row already."""
        from . import fused
        if tuple(fused.PATCH_TILE) == (1, 1):
            return self._fplan, self._fplan_d
        if getattr(self, "_fplan_m", None) is None:
            self._fplan_m = self._build_quad_plan((1, 1))
            self._fplan_m_d = self._fplan_m.to_device(self.device)
        return self._fplan_m, self._fplan_m_d

    def fused_ready(self):
        """True when the fused route can run on this mesh.  The patch plan has limits the two-kernel route does not
        have (a node pair shared by more than 32 elements, a node touching more than a patch holds, more than 255
        blocks stored by one wave): a mesh beyond them is remembered as not fusable and the default entry points fall
        through to element kernel + assembly kernel -- they never raise for it."""
        if not self.can_fuse:
            return False
        if self._fplan_d is None:
            try:
                self.fused_plan()
            except (ValueError, RuntimeError) as exc:     # FemShellHipError is a RuntimeError
                self._fplan_error = exc
                self._fplan = self._fplan_d = None
                return False
        return True

    def assemble_stiffness_fused(self, status=None, mass=False) -> CsrMatrix:
        """K straight from the coordinates in one launch: element evaluation (core/assembler.py:160-193 ->
        MITC4.K) and the ADD_VALUES loop (:287-303) fused, no element matrices in memory.  Bit-identical to
        refresh_element_matrices() + the descriptor assembly.  `status`: optional int32 [E] device tensor.
        mass=True: the consistent mass matrix the same way (FS_FUSED_MASS; MITC4.M, core/assembler.py:305-336)."""
        if not self.can_fuse:
            raise NotImplementedError("the fused kernels cover uniform MITC4+ quad meshes and HEXA8 meshes (full rule)"
                                      + (f" (plan: {self._fplan_error})" if self._fplan_error is not None else ""))
        fp = self.fused_plan()
        d, g = self._fplan_d, self._groups[0]
        lib = _lib.load()
        data = torch.empty(self._plan.nnz, dtype=torch.float64, device=self.device)
        if self._family == ElementFamily.SOLID:
            # HEXA8.K of every element (elements/SOLID.py:543-567) and the ADD_VALUES loop (core/assembler.py:287-303)
            if mass:          # SolidElement.M of every element (elements/SOLID.py:569-597) and the loop of :305-336
                _lib.check(lib.fs_fused_assemble_hex8_mass(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(self._props_d),
                                                           len(self._prop_table), g["E"], fp.n_patches, _lib.ptr(d["pat_eptr"]),
                                                           _lib.ptr(d["pat_elems"]), _lib.ptr(d["pat_nptr"]), _lib.ptr(d["nodes"]),
                                                           _lib.ptr(d["pat_rptr"]), _lib.ptr(d["runs"]), _lib.ptr(d["pat_lptr"]),
                                                           _lib.ptr(d["steps"]), _lib.ptr(data), _lib.stream_ptr()),
                           "fs_fused_assemble_hex8_mass")
                return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))
            if status is not None:
                status.zero_()                     # the kernel ORs FS_ST_BAD_JAC in
            _lib.check(lib.fs_fused_assemble_hex8(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(self._props_d),
                                                  len(self._prop_table), g["E"], fp.n_patches, _lib.ptr(d["pat_eptr"]),
                                                  _lib.ptr(d["pat_elems"]), _lib.ptr(d["pat_nptr"]), _lib.ptr(d["nodes"]),
                                                  _lib.ptr(d["pat_rptr"]), _lib.ptr(d["runs"]), _lib.ptr(d["pat_lptr"]),
                                                  _lib.ptr(d["steps"]), _lib.ptr(data), _lib.ptr(status), _lib.stream_ptr()),
                       "fs_fused_assemble_hex8")
            return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))
        if g["nn"] == 3:
            # MITC3.K / MITC3Composite.K of every element (elements/MITC3.py:935-1103) and the ADD_VALUES loop
            if mass:          # MITC3.M of every element (elements/MITC3.py:1105-1138) and the loop of core/assembler.py:305-336
                _lib.check(lib.fs_fused_assemble_tri_mass(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(self._props_d),
                                                          len(self._prop_table), g["E"], fp.n_patches, _lib.ptr(d["pat_eptr"]),
                                                          _lib.ptr(d["pat_elems"]), _lib.ptr(d["pat_nptr"]), _lib.ptr(d["nodes"]),
                                                          _lib.ptr(d["pat_lptr"]), _lib.ptr(d["steps"]), _lib.ptr(data),
                                                          _lib.stream_ptr()), "fs_fused_assemble_tri_mass")
                return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))
            _lib.check(lib.fs_fused_assemble_tri(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(self._props_d),
                                                 len(self._prop_table), g["E"], fp.n_patches, _lib.ptr(d["pat_eptr"]),
                                                 _lib.ptr(d["pat_elems"]), _lib.ptr(d["pat_nptr"]), _lib.ptr(d["nodes"]),
                                                 _lib.ptr(d["pat_lptr"]), _lib.ptr(d["steps"]), _lib.ptr(data),
                                                 _lib.ptr(status), _lib.stream_ptr()), "fs_fused_assemble_tri")
            return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))
        if mass:
            fp, d = self._mass_plan()        # the same patches, numbered row by row
        _lib.check(lib.fs_fused_assemble_quad(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(self._props_d),
                                              len(self._prop_table), g["E"], fp.n_patches, _lib.ptr(d["pat_eptr"]),
                                              _lib.ptr(d["pat_elems"]), _lib.ptr(d["pat_rptr"]), _lib.ptr(d["recs"]),
                                              fp.max_patch_elems, _lib.FS_FUSED_MASS if mass else self._fused_flags,
                                              _lib.ptr(data), _lib.ptr(status), _lib.stream_ptr()),
                   "fs_fused_assemble_quad")
        return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))

    def assemble_mass_fused(self) -> CsrMatrix:
        return self.assemble_stiffness_fused(mass=True)

    def assemble_stiffness_matrix(self, recompute=False, fused=None) -> CsrMatrix:
        """core/assembler.py:269-303.  recompute=True re-evaluates the element matrices: on uniform MITC4+ meshes
        through the fused kernel (fused=False forces element kernel + assembly kernel: the same sums of the same block
        function, equal to rounding -- see fs_fused_assemble_quad in include/femshell_hip.h)."""
        fused = self.use_fused if fused is None else fused
        if (recompute or self._K_blk is None) and fused and self.fused_ready():
            # the geometry was checked when the assembler was built (the reference checks in its element constructors
            # only); a re-evaluation does not sync with the host again.  The status words of THIS evaluation are
            # written (check_status() reads them); the cached element blocks (`_ke_array`, the linear blocks the
            # Newton path reuses) are dropped, so that they are re-evaluated from the current inputs on next use.
            st = self._status[0] if self._status and self._status[0].numel() == self._groups[0]["E"] else None
            if recompute:
                self._K_blk = None
                self._K_lin_csr = None
            return self.assemble_stiffness_fused(status=st)
        if recompute or self._K_blk is None:
            self._K_lin_csr = None
            self.refresh_element_matrices(want_K=True, want_M=False, check=self._K_blk is None)
        return self._assemble(self._K_blk)

    def assemble_mass_matrix(self, recompute=False, fused=None) -> CsrMatrix:
        """core/assembler.py:305-336.  On uniform MITC4+, MITC3+ and HEXA8 meshes a (re-)evaluation goes through the fused kernels (no element
        mass matrices in memory; `_me_array` still materialises them on demand); fused=False forces element kernel +
        assembly kernel (same bits)."""
        fused = self.use_fused if fused is None else fused
        if (recompute or self._M_blk is None) and fused and self.fused_ready():      # MITC4+, MITC3+ and HEXA8
            return self.assemble_mass_fused()
        if recompute or self._M_blk is None:
            self.refresh_element_matrices(want_K=False, want_M=True, check=False)
        return self._assemble(self._M_blk)

    # --- new (SURVEY.md 8a "New"): Newton re-assembly of K_T(u) and f_int(u) -----------------
    def _element_displacements(self, u):
        """u_e[e] = u[dofs_e] for every element group (fs_gather_dofs)."""
        lib = _lib.load()
        u = torch.as_tensor(u, dtype=torch.float64, device=self.device).contiguous()
        if u.numel() != self.dofs_count:
            raise ValueError(f"Expected {self.dofs_count} DOFs, got {u.numel()}")
        d = self.dofs_per_node
        out = []
        for g in self._groups:
            if "dofs_d" not in g:
                ar = torch.arange(d, dtype=torch.int64, device=self.device)
                g["dofs_d"] = (g["conn_d"].to(torch.int64)[:, :, None] * d + ar).reshape(g["E"], -1).contiguous()
            nd = g["nn"] * d
            ue = torch.empty((g["E"], nd), dtype=torch.float64, device=self.device)
            _lib.check(lib.fs_gather_dofs(_lib.ptr(u), _lib.ptr(g["dofs_d"]), g["E"] * nd, _lib.ptr(ue),
                                          _lib.stream_ptr()), "fs_gather_dofs")
            out.append(ue)
        return out

    def _nl_pass(self, u, want_KT, want_F):
        """Per group: update_configuration(u_e) + compute_tangent_stiffness / compute_internal_forces."""
        if self._family != ElementFamily.SHELL:
            raise NotImplementedError("K_T / f_int re-assembly is available for shell meshes (MITC4 / MITC3)")
        ues = self._element_displacements(u)
        # the linear blocks do not depend on u: the tiled quad path hands the ones of the last K evaluation back to the
        # kernel instead of recomputing them every iterate (the reference reads its cached MITC4.K the same way)
        tiled = self._layout == _lib.FS_LAYOUT_SYMBLK_T16
        if self.reuse_linear_blocks:
            self._k_blocks()
        can = self.reuse_linear_blocks and self._K_blk is not None
        group_cached = [can and (tiled or g["nn"] == 3) for g in self._groups]   # element-major quads recompute
        buf = None
        if want_KT or not all(group_cached):
            buf = torch.empty((self._plan.n_src_blocks, self.dofs_per_node ** 2), dtype=torch.float64, device=self.device)
        fparts = []
        for g, ue, cached in zip(self._groups, ues, group_cached):
            extra = {"K_lin": self._group_view(self._K_blk, g)} if cached else {}
            out = batch.shell_batch(g["nn"], g["xyz_d"], g["pid_d"], self._props_d, len(self._prop_table),
                                    want_K=False, status=False, validate=False, u_e=ue, want_KT=want_KT,
                                    want_F=want_F, KT_out=None if buf is None else self._group_view(buf, g),
                                    layout=self._layout, **extra)
            if want_F:
                fparts.append(out["f_int"].reshape(-1))
        f = None
        if want_F:
            f = fparts[0] if len(fparts) == 1 else torch.cat(fparts)
        return {"KT": buf if want_KT else None, "f_int": f}

    def _scatter_vector(self, fe):
        """fe: flat device buffer of the groups' element vectors, group after group."""
        lib = _lib.load()
        if not hasattr(self, "_vplan_d"):
            seg, src = vector_plan(self._conn, int(self.mesh.node_count), self.dofs_per_node)
            self._vplan_d = (torch.from_numpy(seg).to(self.device), torch.from_numpy(src).to(self.device))
        seg, src = self._vplan_d
        f = torch.empty(self.dofs_count, dtype=torch.float64, device=self.device)
        _lib.check(lib.fs_vec_gather_sum(_lib.ptr(fe), _lib.ptr(src), _lib.ptr(seg), self.dofs_count,
                                         _lib.ptr(f), _lib.stream_ptr()), "fs_vec_gather_sum")
        return f

    # --- the Newton path on the ASSEMBLED linear matrix (uniform MITC4+ meshes) ---------------------
    def _newton_fused(self):
        """K does not depend on u: on uniform MITC4+ meshes with a patch plan the Newton re-assembly works on the
        assembled linear matrix (kept once, 2 592 B per element) instead of streaming 2 880 B of element blocks per
        element through an element kernel and an assembly kernel every iterate."""
        return (self.newton_on_assembled and self.use_fused and self._family == ElementFamily.SHELL
                and len(self._groups) == 1 and self._groups[0]["nn"] == 4 and self._groups[0]["pos"] is None
                and self.fused_ready())

    def _linear_csr(self):
        """Values of the assembled linear K (fs_fused_assemble_quad), evaluated on first use and after a
        recompute=True assembly; reuse_linear_blocks=False re-evaluates them on every call."""
        if self._K_lin_csr is None or not self.reuse_linear_blocks:
            self._K_lin_csr = self.assemble_stiffness_fused().data
        return self._K_lin_csr

    def _u_nodes(self, u):
        u = torch.as_tensor(u, dtype=torch.float64, device=self.device).contiguous()
        if u.numel() != self.dofs_count:
            raise ValueError(f"Expected {self.dofs_count} DOFs, got {u.numel()}")
        return u

    def _linear_packed(self, k_lin):
        """The blocks I <= J of the linear matrix in the order the plan's waves emit them (fs_fused_tangent_pack): what
        fs_fused_assemble_quad_tangent reads, made once per evaluation of the linear matrix."""
        lib, d = _lib.load(), self._fplan_d
        if self._K_lin_packed is None or self._K_lin_packed[0] is not k_lin:
            n_recs = int(d["recs"].numel() if d["recs"].dtype != torch.uint8 else d["recs"].numel() // 64)
            if "wave_off" not in d:
                d["wave_off"] = torch.empty(n_recs // 64 + 1, dtype=torch.int64, device=self.device)
                nb = ctypes.c_int64(0)
                _lib.check(lib.fs_fused_tangent_offsets(n_recs, _lib.ptr(d["recs"]), _lib.ptr(d["wave_off"]),
                                                        ctypes.byref(nb), _lib.stream_ptr()), "fs_fused_tangent_offsets")
                d["n_upper_blocks"] = int(nb.value)
            packed = torch.empty(36 * d["n_upper_blocks"], dtype=torch.float64, device=self.device)
            _lib.check(lib.fs_fused_tangent_pack(n_recs, _lib.ptr(d["recs"]), _lib.ptr(d["wave_off"]), _lib.ptr(k_lin),
                                                 _lib.ptr(packed), _lib.stream_ptr()), "fs_fused_tangent_pack")
            self._K_lin_packed = (k_lin, packed)
        return self._K_lin_packed[1]

    def _tangent_on_assembled(self, u, k_lin, f_nl=None) -> CsrMatrix:
        lib, fp, d, g = _lib.load(), self.fused_plan(), self._fplan_d, self._groups[0]
        if "pat_conn" not in d:       # connectivity rows of the plan's element slots (int32), built once
            d["pat_conn"] = g["conn_d"][d["pat_elems"].to(torch.int64)].to(torch.int32).contiguous()
        packed = self._linear_packed(k_lin)
        data = torch.empty(self._plan.nnz, dtype=torch.float64, device=self.device)
        _lib.check(lib.fs_fused_assemble_quad_tangent(
            _lib.ptr(g["xyz_d"]), _lib.ptr(d["pat_conn"]), _lib.ptr(u), _lib.ptr(g["pid_d"]), _lib.ptr(self._props_d),
            len(self._prop_table), g["E"], fp.n_patches, _lib.ptr(d["pat_eptr"]), _lib.ptr(d["pat_elems"]),
            _lib.ptr(d["pat_rptr"]), _lib.ptr(d["recs"]), fp.max_patch_elems, _lib.ptr(d["wave_off"]), _lib.ptr(packed),
            _lib.ptr(data), _lib.ptr(f_nl), self.tangent_flags, _lib.stream_ptr()), "fs_fused_assemble_quad_tangent")
        return CsrMatrix(self._indptr, self._indices, data, (self.dofs_count, self.dofs_count))

    tangent_flags = 0            # FS_TANGENT_* flags of fs_fused_assemble_quad_tangent

    def _forces_on_assembled(self, u, k_lin, f_nl=None):
        """f_int = K u (assembled matrix) + element-order sums of the nonlinear membrane forces (f_nl: already formed
        by the tangent kernel of the same iterate)."""
        lib, g, pd = _lib.load(), self._groups[0], self._plan_d
        if f_nl is None:
            fe = torch.empty((g["E"], 24), dtype=torch.float64, device=self.device)
            _lib.check(lib.fs_shell_quad_fint_nl(_lib.ptr(g["xyz_d"]), _lib.ptr(g["conn_d"]), _lib.ptr(u), _lib.ptr(g["pid_d"]),
                                                 _lib.ptr(self._props_d), len(self._prop_table), g["E"], _lib.ptr(fe),
                                                 _lib.stream_ptr()), "fs_shell_quad_fint_nl")
            f_nl = self._scatter_vector(fe.reshape(-1))
        f = torch.empty(self.dofs_count, dtype=torch.float64, device=self.device)
        _lib.check(lib.fs_csr_blk_matvec(_lib.ptr(pd["nbr_ptr"]), _lib.ptr(pd["nbr"]), _lib.ptr(pd["blk_row"]),
                                         int(self._plan.n_nodes), self.dofs_per_node, _lib.ptr(k_lin), _lib.ptr(u),
                                         _lib.ptr(f_nl), _lib.ptr(f), _lib.stream_ptr()), "fs_csr_blk_matvec")
        return f

    def assemble_tangent_stiffness(self, u) -> CsrMatrix:
        """K_T(u): per element update_configuration(u[dofs]); compute_tangent_stiffness()
        (elements/MITC4.py:1067-1084,1246-1298), scatter-added like K."""
        if self._newton_fused():
            return self._tangent_on_assembled(self._u_nodes(u), self._linear_csr())
        return self._assemble(self._nl_pass(u, True, False)["KT"])

    def assemble_internal_forces(self, u):
        """f_int(u): per element compute_internal_forces() (elements/MITC4.py:1300-1341),
        scatter-added in element order. Returns a device vector [dofs_count]."""
        if self._newton_fused():
            return self._forces_on_assembled(self._u_nodes(u), self._linear_csr())
        return self._scatter_vector(self._nl_pass(u, False, True)["f_int"])

    def assemble_tangent_and_forces(self, u):
        if self._newton_fused():
            u, k_lin = self._u_nodes(u), self._linear_csr()
            f_nl = torch.zeros(self.dofs_count, dtype=torch.float64, device=self.device)     # (nodes without elements stay 0)
            KT = self._tangent_on_assembled(u, k_lin, f_nl)
            return KT, self._forces_on_assembled(u, k_lin, f_nl)
        out = self._nl_pass(u, True, True)
        return self._assemble(out["KT"]), self._scatter_vector(out["f_int"])

    def capture_tangent_and_forces(self):
        """One HIP graph for a whole Newton re-assembly (gather u_e -> element kernels -> CSR assembly -> vector
        scatter): small meshes are launch-bound (five launches of a few microseconds each), the graph replays them
        with one submission.  Returns step(u) -> (K_T, f_int); every call overwrites the SAME output buffers."""
        u_static = torch.zeros(self.dofs_count, dtype=torch.float64, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up outside the capture: lazy plans, allocator pools
            self.assemble_tangent_and_forces(u_static)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            K, f = self.assemble_tangent_and_forces(u_static)

        def step(u):
            u_static.copy_(torch.as_tensor(u, dtype=torch.float64, device=self.device))
            graph.replay()
            return K, f

        step.graph = graph
        return step

    def assemble_load_vector(self, load_condition):
        """core/assembler.py:338-384: fe = element.body_load(load_condition.value), scatter-added in
        element order.  Shells have no body_load in the reference (AttributeError)."""
        if self._family != ElementFamily.SOLID:
            name = "MITC3" if self._groups[0]["nn"] == 3 else "MITC4"
            raise AttributeError(f"'{name}' object has no attribute 'body_load'")
        value = getattr(load_condition, "value", load_condition)
        parts = []
        for g in self._groups:
            out = batch.solid_batch(g["nn"], g["xyz_d"], g["pid_d"], self._props_d, len(self._prop_table),
                                    want_K=False, want_M=False, body=value, reduced=self._reduced,
                                    status=False, validate=False)
            parts.append(out["f"].reshape(-1))
        return self._scatter_vector(parts[0] if len(parts) == 1 else torch.cat(parts))

    def assemble_geometric_stiffness(self, stress_field=None, omega=None, rotation_axis=None,
                                     rotation_center=None) -> CsrMatrix:
        """core/assembler.py:386-516.  Per element: sigma = stress_field[element id] if present, else
        compute_centrifugal_prestress(omega, axis, center) (MITC3.py:1633-1715, SOLID.py:268-320), else
        skipped; elements with max|sigma| < 1e-20 are skipped; K_G^e = compute_geometric_stiffness(sigma)
        (MITC4.py:1191-1225, MITC3.py:1501-1580, SOLID.py:322-374), scatter-added like K.
        `stress_field`: {element id: stress} or an array in element order -- shells [sxx,syy,sxy] in
        element-local axes (or 2x2), solids Voigt [xx,yy,zz,xy,yz,zx].  Like the reference, the omega
        mode raises AttributeError for MITC4 (it has no compute_centrifugal_prestress).  Unlike the
        PETSc matrix of the reference the returned CSR keeps K's full pattern (explicit zeros)."""
        if stress_field is None and omega is None:
            raise ValueError("Either 'stress_field' or 'omega' must be provided for geometric stiffness")
        axis = np.array([0.0, 0.0, 1.0]) if rotation_axis is None else np.asarray(rotation_axis, dtype=float)
        axis = axis / np.linalg.norm(axis)
        center = np.zeros(3) if rotation_center is None else np.asarray(rotation_center, dtype=float)
        solid = self._family == ElementFamily.SOLID
        ns = 6 if solid else 3
        E = self._conn.shape[0]
        given = np.zeros(E, dtype=bool)
        sig = np.zeros((E, ns))
        if isinstance(stress_field, dict):
            for eid, sv in stress_field.items():
                if eid in self._elem_pos:
                    sv = np.asarray(sv, dtype=float)
                    if not solid and sv.shape == (2, 2):
                        sv = np.array([sv[0, 0], sv[1, 1], sv[0, 1]])
                    sig[self._elem_pos[eid]] = sv
                    given[self._elem_pos[eid]] = True
        elif stress_field is not None:
            sig = np.ascontiguousarray(stress_field, dtype=np.float64).reshape(E, ns).copy()
            given[:] = True
        if solid:   # the shell kernel applies the 1e-20 skip rule itself
            sig = np.where(np.abs(sig).max(axis=1, keepdims=True) < 1e-20, 0.0, sig)
        buf = torch.empty((self._plan.n_src_blocks, self.dofs_per_node ** 2), dtype=torch.float64, device=self.device)
        for g in self._groups:
            sel = slice(None) if g["pos"] is None else g["pos"]
            sg = torch.from_numpy(np.ascontiguousarray(sig[sel])).to(self.device)
            gv = given[sel]
            if omega is not None and not gv.all():
                cf = batch.centrifugal_prestress(solid, g["nn"], g["xyz_d"], g["pid_d"], self._props_d,
                                                 len(self._prop_table), omega, axis, center)
                if solid:
                    cf = torch.where(cf.abs().amax(dim=1, keepdim=True) < 1e-20, torch.zeros_like(cf), cf)
                sg = torch.where(torch.from_numpy(gv).to(self.device)[:, None], sg, cf)
            if solid:
                out = batch.solid_batch(g["nn"], g["xyz_d"], g["pid_d"], self._props_d, len(self._prop_table),
                                        want_K=False, want_M=False, sigma=sg, reduced=self._reduced,
                                        status=False, validate=False)
                self._group_view(buf, g).copy_(out["KG"])
            else:
                batch.shell_kg_batch(g["nn"], g["xyz_d"], g["pid_d"], self._props_d, len(self._prop_table), sg,
                                     KG_out=self._group_view(buf, g), layout=self._layout)
        return self._assemble(buf)


class _RangePos:
    """element id -> position for ArrayMesh-sized meshes without building a 10M-entry dict."""

    def __init__(self, n):
        self.n = n

    def __getitem__(self, k):
        if not 0 <= k < self.n:
            raise KeyError(k)
        return k

    def __contains__(self, k):
        return 0 <= k < self.n
