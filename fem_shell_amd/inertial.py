"""Inertial (fictitious) nodal forces of a rotating frame on the device -- the rotor load that goes with the lumped
mass of the assembled M.  Mirrors InertialForcesCalculator of the reference
(solvers/fsi/corotational.py:304-562): same constructor, same methods, same diagnostics; inputs may be NumPy arrays or
device tensors, results are device tensors [n_nodes, 3] (fs_inertial_forces)."""
import numpy as np
import torch

from . import _lib, batch


class InertialForcesCalculator:
    def __init__(self, rotation_axis, rotation_center=None):
        axis = np.asarray(rotation_axis, dtype=np.float64)
        norm = np.linalg.norm(axis)
        if norm < 1e-12:
            raise ValueError("rotation_axis cannot be zero vector")
        self._axis = np.ascontiguousarray(axis / norm)
        self._center = np.zeros(3) if rotation_center is None else np.ascontiguousarray(
            np.asarray(rotation_center, dtype=np.float64))

    axis = property(lambda self: self._axis.copy())
    center = property(lambda self: self._center.copy())

    @staticmethod
    def _dev(a, shape_last=None):
        t = torch.as_tensor(a, dtype=torch.float64, device=batch._dev()).contiguous()
        if shape_last is not None and (t.ndim != 2 or t.shape[1] != shape_last):
            raise ValueError(f"expected an array of shape (n_nodes, {shape_last}), got {tuple(t.shape)}")
        return t

    def _run(self, coords, vel, masses, omega, alpha, parts):
        x = self._dev(coords, 3)
        m = self._dev(masses).reshape(-1)
        if m.numel() != x.shape[0]:
            raise ValueError("nodal_masses must hold one value per node")
        v = None if vel is None else self._dev(vel, 3)
        F = torch.empty_like(x)
        _lib.check(_lib.load().fs_inertial_forces(_lib.ptr(x), None if v is None else _lib.ptr(v), _lib.ptr(m), x.shape[0],
                                                  float(omega), float(alpha), self._axis.ctypes.data,
                                                  self._center.ctypes.data, parts, _lib.ptr(F), _lib.stream_ptr()),
                   "fs_inertial_forces")
        return F

    def compute_centrifugal_force(self, nodal_coords, nodal_masses, omega):
        """F = m w^2 r_perp (corotational.py:347-388)."""
        return self._run(nodal_coords, None, nodal_masses, omega, 0.0, _lib.FS_INERTIAL_CENTRIFUGAL)

    def compute_coriolis_force(self, nodal_velocities, nodal_masses, omega):
        """F = -2 m (w x v) (corotational.py:390-429)."""
        return self._run(nodal_velocities, nodal_velocities, nodal_masses, omega, 0.0, _lib.FS_INERTIAL_CORIOLIS)

    def compute_euler_force(self, nodal_coords, nodal_masses, alpha):
        """F = -m (alpha x r) (corotational.py:431-469)."""
        return self._run(nodal_coords, None, nodal_masses, 0.0, alpha, _lib.FS_INERTIAL_EULER)

    def compute_all_inertial_forces(self, nodal_coords, nodal_velocities, nodal_masses, omega, alpha=0.0,
                                    include_centrifugal=True, include_coriolis=True, include_euler=True):
        """Sum of the requested parts and the reference's diagnostics dictionary (corotational.py:471-552)."""
        diagnostics = {}

        def stats(F):
            mag = torch.linalg.norm(F, dim=1)
            return {"total": float(torch.linalg.norm(F.sum(dim=0))), "max_nodal": float(mag.max()),
                    "mean_nodal": float(mag.mean())}

        parts = 0
        if include_centrifugal:
            parts |= _lib.FS_INERTIAL_CENTRIFUGAL
            diagnostics["centrifugal"] = stats(self.compute_centrifugal_force(nodal_coords, nodal_masses, omega))
        if include_coriolis:
            parts |= _lib.FS_INERTIAL_CORIOLIS
            diagnostics["coriolis"] = stats(self.compute_coriolis_force(nodal_velocities, nodal_masses, omega))
        if include_euler:
            parts |= _lib.FS_INERTIAL_EULER
            diagnostics["euler"] = stats(self.compute_euler_force(nodal_coords, nodal_masses, alpha))
        if parts:
            F = self._run(nodal_coords, nodal_velocities if include_coriolis else None, nodal_masses, omega, alpha, parts)
        else:
            F = torch.zeros_like(self._dev(nodal_coords, 3))
        diagnostics["total_inertial"] = stats(F)
        return F, diagnostics
