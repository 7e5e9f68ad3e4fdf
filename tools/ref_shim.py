"""Import shim for the reference's hot-path modules (oracle tooling; runs ONLY in the build
container where /root/reference exists -- never on the GPU box, never from the product path).

Bypasses the two package __init__ files that pull pyvista/gmsh/petsc4py so that
fem_shell.elements / core.laminate / core.material / core.properties import with NumPy alone.
"""
import os
import sys
import types

REF = "/root/reference/src/fem_shell"


def install():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: golden fixtures can only be generated "
                           "in the build container")
    if "fem_shell" in sys.modules and getattr(sys.modules["fem_shell"], "_shimmed", False):
        return

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m._shimmed = True
        sys.modules[name] = m

    ns("fem_shell", REF)
    ns("fem_shell.core", REF + "/core")
    ns("fem_shell.constitutive", REF + "/constitutive")
    mesh = types.ModuleType("fem_shell.core.mesh")  # real one needs pyvista/gmsh/meshio
    mesh.MeshElement = type("MeshElement", (), {})
    mesh.MeshModel = type("MeshModel", (), {})
    sys.modules["fem_shell.core.mesh"] = mesh


def ns_extra(name, path):
    """Further namespace package of the reference (e.g. fem_shell.postprocess) without its __init__."""
    m = types.ModuleType(name)
    m.__path__ = [path]
    m._shimmed = True
    sys.modules[name] = m


def install_assembler_backend():
    """Fake mpi4py / petsc4py so that the UNMODIFIED fem_shell.core.assembler imports
    and runs: Mat.setValuesLocal records (rows x cols, values) and assemble() sums duplicates in
    insertion order into CSR with sorted columns -- the AIJ semantics of MatSetValues(ADD_VALUES) +
    MatAssembly without MAT_IGNORE_ZERO_ENTRIES (explicit zeros are kept)."""
    import numpy as np

    install()
    if "petsc4py" in sys.modules and getattr(sys.modules["petsc4py"], "_shimmed", False):
        return

    class Mat:
        class Option:
            NEW_NONZERO_ALLOCATION_ERR = 0

        def create(self, comm=None):
            self._r, self._c, self._v = [], [], []
            return self

        def setType(self, t): pass

        def setSizes(self, s):
            self.shape = tuple(s)

        def setPreallocationNNZ(self, nnz):
            self.d_nnz = np.asarray(nnz[0]).copy()

        def setUp(self): pass

        def setOption(self, o, v): pass

        def setValuesLocal(self, rows, cols, vals, addv=None):
            rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
            self._r.append(np.repeat(rows, cols.size))
            self._c.append(np.tile(cols, rows.size))
            self._v.append(np.asarray(vals, np.float64).ravel())

        def assemble(self):
            n = self.shape[0]
            if not self._r:
                self.indptr = np.zeros(n + 1, np.int64); self.indices = np.zeros(0, np.int32); self.data = np.zeros(0)
                return
            r, c, v = np.concatenate(self._r), np.concatenate(self._c), np.concatenate(self._v)
            key = r * self.shape[1] + c
            order = np.argsort(key, kind="stable")          # insertion order kept inside one (i, j)
            key, v = key[order], v[order]
            first = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
            cnt = np.diff(np.r_[first, key.size])
            data = np.zeros(first.size)
            for k in range(int(cnt.max())):                 # value = ((v0 + v1) + v2) + ...
                has = cnt > k
                data[has] = data[has] + v[first[has] + k]
            ukey = key[first]
            rows = ukey // self.shape[1]
            self.indices = (ukey % self.shape[1]).astype(np.int32)
            self.indptr = np.r_[0, np.cumsum(np.bincount(rows, minlength=n))].astype(np.int64)
            self.data = data

    class Vec:
        def create(self, comm=None): return self

        def setSizes(self, n):
            self.array = np.zeros(int(n))

        def setUp(self): pass

        def zeroEntries(self):
            self.array[:] = 0.0

        def setValuesLocal(self, idx, vals, addv=None):
            np.add.at(self.array, np.asarray(idx, np.int64), np.asarray(vals, np.float64))

        def assemble(self): pass

    petsc = types.ModuleType("petsc4py")
    petsc._shimmed = True
    P = types.ModuleType("petsc4py.PETSc")
    P.IntType, P.ScalarType = np.int32, np.float64
    P.InsertMode = types.SimpleNamespace(ADD_VALUES=1)
    P.Mat, P.Vec = Mat, Vec
    petsc.PETSc = P
    mpi = types.ModuleType("mpi4py")
    M = types.ModuleType("mpi4py.MPI")
    M.COMM_WORLD = None
    mpi.MPI = M
    sys.modules.update({"petsc4py": petsc, "petsc4py.PETSc": P, "mpi4py": mpi, "mpi4py.MPI": M})
