"""Dirichlet boundary conditions on the device CSR: the reference's `DirichletCondition` /
`BoundaryConditionManager` (core/bc.py:30-290) with the PETSc calls replaced by the HIP kernels of
femshell_sparse.hip (fs_csr_zero_rows, fs_csr_reduce_*, fs_expand_solution)."""
import ctypes
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .assembler import CsrMatrix


class DirichletCondition:
    """core/bc.py:30-55."""

    def __init__(self, dofs: Iterable[int], value: float):
        self.dofs = tuple(sorted(set(int(d) for d in dofs)))
        self.value = value


class BoundaryConditionManager:
    """Same protocol as core/bc.py:57-290 on `CsrMatrix` / device vectors: apply_dirichlet(),
    reduced_system, expand_solution(), fixed_dofs, free_dofs.  Like PETSc's Mat.zeroRows the
    stiffness (and mass) passed in are modified in place; the load vector is copied."""

    def __init__(self, stiffness: CsrMatrix, load, mass: Optional[CsrMatrix] = None, dof_per_node: int = 2):
        if stiffness.shape[0] != stiffness.shape[1]:
            raise ValueError("Stiffness matrix must be square")
        self.K = stiffness
        dev = stiffness.data.device
        self.F = torch.as_tensor(load, dtype=torch.float64, device=dev).clone().contiguous()
        if stiffness.shape[0] != self.F.numel():
            raise ValueError("Stiffness and load dimensions mismatch")
        if mass is not None and tuple(mass.shape) != tuple(stiffness.shape):
            raise ValueError("Mass matrix must match stiffness dimensions")
        self.M = mass
        self.dof_per_node = dof_per_node
        self.n_dof = int(stiffness.shape[0])
        self._fixed_dofs: Dict[int, float] = {}
        self._free_rank = None

    def apply_dirichlet(self, conditions: Iterable[DirichletCondition]) -> None:
        """core/bc.py:116-175."""
        fixed = {}
        for bc in conditions:
            for dof in bc.dofs:
                if not 0 <= dof < self.n_dof:
                    raise ValueError(f"DOF {dof} out of range [0, {self.n_dof - 1}]")
                if dof in fixed and not np.isclose(fixed[dof], bc.value):
                    raise ValueError(f"Conflicting values for DOF {dof}: {fixed[dof]} vs {bc.value}")
                fixed[dof] = bc.value
        self._fixed_dofs = fixed
        lib, dev = _lib.load(), self.K.data.device
        idx = np.fromiter(fixed.keys(), dtype=np.int64, count=len(fixed))
        val = np.fromiter(fixed.values(), dtype=np.float64, count=len(fixed))
        mask = np.zeros(self.n_dof, dtype=np.uint8)
        mask[idx] = 1
        ufix = np.zeros(self.n_dof)
        ufix[idx] = val
        self._mask_d = torch.from_numpy(mask).to(dev)
        self._ufix_d = torch.from_numpy(ufix).to(dev)
        rows = torch.from_numpy(idx).to(dev)
        s = _lib.stream_ptr()
        # K.zeroRows(fixed, 1.0, U_fixed, F) ; M.zeroRows(fixed, 0.0)
        _lib.check(lib.fs_csr_zero_rows(_lib.ptr(self.K.indptr), _lib.ptr(self.K.indices), _lib.ptr(self.K.data),
                                        _lib.ptr(rows), len(idx), 1.0, _lib.ptr(self._ufix_d), _lib.ptr(self.F), s),
                   "fs_csr_zero_rows")
        if self.M is not None:
            _lib.check(lib.fs_csr_zero_rows(_lib.ptr(self.M.indptr), _lib.ptr(self.M.indices), _lib.ptr(self.M.data),
                                            _lib.ptr(rows), len(idx), 0.0, None, None, s), "fs_csr_zero_rows")
        # index sets of the reduced system (core/bc.py:148-155)
        self._free_rank = torch.empty(self.n_dof, dtype=torch.int32, device=dev)
        self._red_indptr = torch.empty(self.n_dof + 1, dtype=torch.int64, device=dev)
        nf, nz = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.fs_csr_reduce_symbolic(_lib.ptr(self.K.indptr), _lib.ptr(self.K.indices), self.n_dof,
                                              _lib.ptr(self._mask_d), _lib.ptr(self._free_rank),
                                              _lib.ptr(self._red_indptr), ctypes.byref(nf), ctypes.byref(nz), s),
                   "fs_csr_reduce_symbolic")
        self._n_free, self._nnz_red = int(nf.value), int(nz.value)
        self._red_indptr = self._red_indptr[:self._n_free + 1]
        self._red_indices = None

    @property
    def fixed_dofs(self) -> Dict[int, float]:
        return self._fixed_dofs.copy()

    @property
    def free_dofs(self) -> np.ndarray:
        if self._free_rank is None:
            raise RuntimeError("Boundary conditions not applied")
        return torch.nonzero(self._free_rank >= 0).reshape(-1).cpu().numpy()

    def _reduce(self, A: CsrMatrix, with_load: bool):
        lib, dev = _lib.load(), A.data.device
        if self._red_indices is None:
            self._red_indices = torch.empty(self._nnz_red, dtype=torch.int32, device=dev)
            ix = self._red_indices
        else:
            ix = None        # K and M share the pattern: written once
        data = torch.empty(self._nnz_red, dtype=torch.float64, device=dev)
        f_red = torch.empty(self._n_free, dtype=torch.float64, device=dev) if with_load else None
        _lib.check(lib.fs_csr_reduce_fill(_lib.ptr(A.indptr), _lib.ptr(A.indices), _lib.ptr(A.data), self.n_dof,
                                          _lib.ptr(self._free_rank), _lib.ptr(self._red_indptr), _lib.ptr(ix),
                                          _lib.ptr(data), _lib.ptr(self._ufix_d) if with_load else None,
                                          _lib.ptr(self.F) if with_load else None, _lib.ptr(f_red),
                                          _lib.stream_ptr()), "fs_csr_reduce_fill")
        return CsrMatrix(self._red_indptr, self._red_indices, data, (self._n_free, self._n_free)), f_red

    @property
    def reduced_system(self) -> Tuple[CsrMatrix, torch.Tensor, Optional[CsrMatrix]]:
        """(K_red, F_red, M_red): free x free sub-matrices and F_free - K_free,fixed u_fixed
        (core/bc.py:178-239)."""
        if self._free_rank is None:
            raise RuntimeError("Boundary conditions not applied")
        if self.M is not None and not (self.M.indptr is self.K.indptr or torch.equal(self.M.indptr, self.K.indptr)):
            raise ValueError("K and M must share one sparsity pattern")
        K_red, F_red = self._reduce(self.K, True)
        M_red = self._reduce(self.M, False)[0] if self.M is not None else None
        return K_red, F_red, M_red

    def expand_solution(self, u_red) -> torch.Tensor:
        """core/bc.py:241-290: full-length vector with the prescribed values at the fixed DOFs."""
        if self._free_rank is None:
            raise RuntimeError("Boundary conditions not applied")
        dev = self.K.data.device
        u_red = torch.as_tensor(u_red, dtype=torch.float64, device=dev).contiguous()
        if u_red.numel() != self._n_free:
            raise ValueError(f"Expected {self._n_free} free DOFs, got {u_red.numel()}")
        out = torch.empty(self.n_dof, dtype=torch.float64, device=dev)
        _lib.check(_lib.load().fs_expand_solution(_lib.ptr(u_red), _lib.ptr(self._free_rank), _lib.ptr(self._ufix_d),
                                                  self.n_dof, _lib.ptr(out), _lib.stream_ptr()), "fs_expand_solution")
        return out

    def expand_state_vector(self, v_red) -> torch.Tensor:
        """Velocity / acceleration: zeros at the fixed DOFs (core/bc.py expand_state_vector)."""
        dev = self.K.data.device
        v_red = torch.as_tensor(v_red, dtype=torch.float64, device=dev).contiguous()
        out = torch.empty(self.n_dof, dtype=torch.float64, device=dev)
        _lib.check(_lib.load().fs_expand_solution(_lib.ptr(v_red), _lib.ptr(self._free_rank), None, self.n_dof,
                                                  _lib.ptr(out), _lib.stream_ptr()), "fs_expand_solution")
        return out


def lump_mass_matrix(M: CsrMatrix) -> torch.Tensor:
    """Row-sum lumping (solvers/fsi/linear_dynamic.py:227-234): the diagonal of the lumped matrix."""
    out = torch.empty(M.shape[0], dtype=torch.float64, device=M.data.device)
    _lib.check(_lib.load().fs_csr_row_sums(_lib.ptr(M.indptr), _lib.ptr(M.data), M.shape[0], _lib.ptr(out),
                                           _lib.stream_ptr()), "fs_csr_row_sums")
    return out
