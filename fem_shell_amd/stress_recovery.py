"""StressRecovery: the reference's post-processing interface (postprocess/stress_recovery.py:110-1542)
on top of batched HIP kernels (csrc/femshell_stress.hip).

Same enums (`StressLocation`, `StressType`), result containers (`StressResult`, `StrainResult`,
`to_dict()` keys) and methods (`compute_element_stresses`, `compute_nodal_stresses`,
`compute_nodal_stresses_all_layers[_dict]`, `compute_element_strains`, `compute_nodal_strains`) and the
module-level helpers (`compute_von_mises`, `compute_principal_stresses`, `compute_max_shear_stress`).
`domain` is a fem_shell_amd MeshAssembler; `u` the full displacement vector (NumPy, torch, or anything
with `.array` like a PETSc Vec).  Per call: gather u_e on the device -> B u -> C eps per element point
-> node-by-node weighted average in element order -> invariants; results come back as NumPy arrays.

Reference behaviour kept on purpose (the parity fixtures tests/golden/stress_ref.npz pin it):
  * shells: B_m / B_kappa act on the global-axis element vector without a T-transformation
    (:497-524, 588-595);
  * shells: the triplet [xx, yy, xy] is stored in Voigt columns 0..2 (:759, :845) while the result
    object reads tau_xy from column 3 (:1279, :1286) -- so `sigma_xy` is 0, `von_mises` sees the shear
    in the zz slot and the Mohr circle runs on (xx, yy, 0);
  * solids in a model: is_3d results (angle 0); the Cardano form takes cos(3 phi) = -q / (2 (p/3)^1.5)
    (:1216-1230), whose roots are the true principal stresses mirrored about the mean stress.
"""
from dataclasses import dataclass
from enum import Enum
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .elements import ElementFamily
from .symbolic import vector_plan


class StressLocation(Enum):
    TOP = "top"
    MIDDLE = "middle"
    BOTTOM = "bottom"


class StressType(Enum):
    MEMBRANE = "membrane"
    BENDING = "bending"
    TOTAL = "total"


@dataclass
class StressResult:
    sigma_xx: np.ndarray
    sigma_yy: np.ndarray
    sigma_xy: np.ndarray
    von_mises: np.ndarray
    sigma_1: np.ndarray
    sigma_2: np.ndarray
    tau_max: np.ndarray
    principal_angle: np.ndarray
    sigma_zz: Optional[np.ndarray] = None
    tau_yz: Optional[np.ndarray] = None
    tau_zx: Optional[np.ndarray] = None
    sigma_3: Optional[np.ndarray] = None

    def to_dict(self) -> Dict[str, np.ndarray]:
        d = {"sigma_xx": self.sigma_xx, "sigma_yy": self.sigma_yy, "sigma_xy": self.sigma_xy,
             "von_mises": self.von_mises, "sigma_1": self.sigma_1, "sigma_2": self.sigma_2,
             "tau_max": self.tau_max, "principal_angle": np.degrees(self.principal_angle)}
        for k in ("sigma_zz", "tau_yz", "tau_zx", "sigma_3"):
            if getattr(self, k) is not None:
                d[k] = getattr(self, k)
        return d


@dataclass
class StrainResult:
    epsilon_xx: np.ndarray
    epsilon_yy: np.ndarray
    gamma_xy: np.ndarray
    epsilon_1: np.ndarray
    epsilon_2: np.ndarray
    gamma_max: np.ndarray
    epsilon_zz: Optional[np.ndarray] = None
    gamma_yz: Optional[np.ndarray] = None
    gamma_zx: Optional[np.ndarray] = None
    epsilon_3: Optional[np.ndarray] = None

    def to_dict(self) -> Dict[str, np.ndarray]:
        d = {"epsilon_xx": self.epsilon_xx, "epsilon_yy": self.epsilon_yy, "gamma_xy": self.gamma_xy,
             "epsilon_1": self.epsilon_1, "epsilon_2": self.epsilon_2, "gamma_max": self.gamma_max}
        for k in ("epsilon_zz", "gamma_yz", "gamma_zx", "epsilon_3"):
            if getattr(self, k) is not None:
                d[k] = getattr(self, k)
        return d


_LOC = {StressLocation.TOP: 1, StressLocation.MIDDLE: 0, StressLocation.BOTTOM: -1}
_PARTS = {StressType.MEMBRANE: _lib.FS_STRESS_MEMBRANE, StressType.BENDING: _lib.FS_STRESS_BENDING,
          StressType.TOTAL: _lib.FS_STRESS_MEMBRANE | _lib.FS_STRESS_BENDING}
_GP = 1.0 / np.sqrt(3.0)


def _solid_rule(nn, reduced):
    """Integration points of the linear solids (SOLID.py:731-754, 1099-1127, 1575-1579, 1744-1766)."""
    if nn == 8:
        if reduced:
            return np.zeros((1, 3))
        return np.array([[(_GP if q & 1 else -_GP), (_GP if q & 2 else -_GP), (_GP if q & 4 else -_GP)]
                         for q in range(8)])
    if nn == 4:
        return np.full((1, 3), 0.25)
    if nn == 6:
        tp = [(1 / 6, 1 / 6), (2 / 3, 1 / 6), (1 / 6, 2 / 3)]
        return np.array([[tp[q // 2][0], tp[q // 2][1], (_GP if q & 1 else -_GP)] for q in range(6)])
    a, za = 0.26318405556971359557, 0.54415184401122528880
    b, zb = 0.50661630334978742377, 0.12251482265544137787
    sx, sy = (-1, 1, 1, -1), (-1, -1, 1, 1)
    return np.array([[sx[q] * a, sy[q] * a, za] for q in range(4)] + [[sx[q] * b, sy[q] * b, zb] for q in range(4)])


def _solid_N(nn, xi, eta, zeta):
    """Shape functions of the linear solids (SOLID.py:756-790, 1129-1160, 1581-1586, 1768-1790)."""
    if nn == 8:
        s = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
        return np.array([0.125 * (1 + a * xi) * (1 + b * eta) * (1 + c * zeta) for a, b, c in s])
    if nn == 4:
        return np.array([1.0 - xi - eta - zeta, xi, eta, zeta])
    if nn == 6:
        L = (1.0 - xi - eta, xi, eta)
        return np.array([l * 0.5 * (1 - zeta) for l in L] + [l * 0.5 * (1 + zeta) for l in L])
    r = max(1.0 - zeta, 1e-14)
    return np.array([0.25 * (r - xi) * (r - eta) / r, 0.25 * (r + xi) * (r - eta) / r,
                     0.25 * (r + xi) * (r + eta) / r, 0.25 * (r - xi) * (r + eta) / r, zeta])


# parametric node positions of the shells (stress_recovery.py:355-358)
_SHELL_NODE_COORDS = {3: [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)],
                      4: [(-1.0, -1.0), (1.0, -1.0), (1.0, 1.0), (-1.0, 1.0)]}


class StressRecovery:
    """stress_recovery.py:280-1402."""

    def __init__(self, domain, u):
        self.domain = domain
        if hasattr(u, "array"):
            u = u.array
        dev = domain.device
        if isinstance(u, torch.Tensor):
            self._u = u.detach().to(device=dev, dtype=torch.float64).reshape(-1).clone()
        else:
            self._u = torch.from_numpy(np.ascontiguousarray(np.asarray(u, dtype=np.float64)).reshape(-1).copy()).to(dev)
        self.u = self._u.cpu().numpy()
        self.dofs_per_node = domain.dofs_per_node
        self.n_nodes = int(domain.mesh.node_count)
        self.n_elements = len(domain._element_map)
        self._solid = domain._family == ElementFamily.SOLID
        self._extrap_cache: Dict[int, np.ndarray] = {}
        self._ue = None
        self._node_plan = None

    # ------------------------------------------------------------------ plumbing
    def _element_u(self):
        if self._ue is None:
            self._ue = self.domain._element_displacements(self._u)
        return self._ue

    def _plan(self):
        """node -> (element, local node) sources in element order, over the groups' [E_g, nn] buffers."""
        if self._node_plan is None:
            seg, src = vector_plan(self.domain._conn, self.n_nodes, 1)
            dev = self.domain.device
            self._node_plan = (torch.from_numpy(seg).to(dev), torch.from_numpy(src).to(dev))
        return self._node_plan

    def _get_extrapolation_matrix(self, nn) -> np.ndarray:
        """E = inv(N_gp) (n_gp == n_nodes) or pinv(N_gp), [nn, n_gp] (stress_recovery.py:437-492)."""
        if nn not in self._extrap_cache:
            if nn > 8:      # quadratic solids: the library's host table (bit-identical to the reference's N)
                import ctypes
                ng = ctypes.c_int32(0)
                v = np.zeros(27 * nn * 4)
                _lib.check(_lib.load().fs_solid2_table(nn, None, None, v.ctypes.data, ctypes.byref(ng)), "fs_solid2_table")
                N_gp = v[:ng.value * nn * 4].reshape(ng.value, nn, 4)[:, :, 0].copy()
            else:
                pts = _solid_rule(nn, self.domain._reduced and nn == 8)
                N_gp = np.array([_solid_N(nn, *p) for p in pts])
            self._extrap_cache[nn] = np.linalg.inv(N_gp) if N_gp.shape[0] == nn else np.linalg.pinv(N_gp)
        return self._extrap_cache[nn]

    def _shell_points(self, pts, location, parts, strain, want_weight):
        """Per group: values [E_g, npts, 3] (+ weights [E_g, npts])."""
        lib, a = _lib.load(), self.domain
        out = []
        for g, ue in zip(a._groups, self._element_u()):
            nn, E = g["nn"], g["E"]
            p = np.ascontiguousarray(np.asarray(pts[nn] if isinstance(pts, dict) else pts, dtype=np.float64).reshape(-1, 2))
            vals = torch.empty((E, p.shape[0], 3), dtype=torch.float64, device=a.device)
            w = torch.empty((E, p.shape[0]), dtype=torch.float64, device=a.device) if want_weight else None
            _lib.check(lib.fs_shell_stress_batch(nn, _lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(a._props_d),
                                                 len(a._prop_table), _lib.ptr(ue), E, p.ctypes.data, p.shape[0],
                                                 _LOC[location], parts, int(strain), _lib.ptr(vals), _lib.ptr(w),
                                                 _lib.stream_ptr()), "fs_shell_stress_batch")
            out.append((vals, w))
        return out

    def _solid_values(self, mode, strain, want_weight):
        lib, a = _lib.load(), self.domain
        out = []
        for g, ue in zip(a._groups, self._element_u()):
            nn, E = g["nn"], g["E"]
            rows = {_lib.FS_SOLID_STRESS_MEAN: 1, _lib.FS_SOLID_STRESS_NODES: nn}[mode]
            vals = torch.empty((E, rows, 6), dtype=torch.float64, device=a.device)
            w = torch.empty(E, dtype=torch.float64, device=a.device) if want_weight else None
            ex = None
            if mode == _lib.FS_SOLID_STRESS_NODES:
                ex = torch.from_numpy(np.ascontiguousarray(self._get_extrapolation_matrix(nn))).to(a.device)
            _lib.check(lib.fs_solid_stress_batch(nn, int(a._reduced), _lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]),
                                                 _lib.ptr(a._props_d), len(a._prop_table), _lib.ptr(ue), E,
                                                 _lib.ptr(ex), mode, None if strain else _lib.ptr(vals),
                                                 _lib.ptr(vals) if strain else None, _lib.ptr(w), _lib.stream_ptr()),
                       "fs_solid_stress_batch")
            if w is not None:
                w = w[:, None].expand(E, rows).contiguous()
            out.append((vals, w))
        return out

    def _element_rows(self, per_group):
        """[n_elements, 6] device array in mesh element order from per-group [E_g, 1, ncomp] values."""
        a = self.domain
        full = torch.zeros((a._conn.shape[0], 6), dtype=torch.float64, device=a.device)
        for g, (vals, _) in zip(a._groups, per_group):
            v = vals.reshape(g["E"], -1)
            if g["pos"] is None:
                full[:, :v.shape[1]] = v
            else:
                full[torch.from_numpy(np.asarray(g["pos"])).to(a.device), :v.shape[1]] = v
        return full

    def _nodal(self, per_group, ncomp, weighted):
        lib, a = _lib.load(), self.domain
        vals = torch.cat([v.reshape(-1, ncomp) for v, _ in per_group]) if len(per_group) > 1 else per_group[0][0].reshape(-1, ncomp)
        w = None
        if weighted:
            w = torch.cat([x.reshape(-1) for _, x in per_group]) if len(per_group) > 1 else per_group[0][1].reshape(-1)
        seg, src = self._plan()
        out = torch.empty((self.n_nodes, 6), dtype=torch.float64, device=a.device)
        _lib.check(lib.fs_nodal_average(_lib.ptr(vals.contiguous()), ncomp, _lib.ptr(w), _lib.ptr(seg), _lib.ptr(src),
                                        self.n_nodes, _lib.ptr(out), _lib.stream_ptr()), "fs_nodal_average")
        return out

    def _invariants(self, voigt, strain):
        lib = _lib.load()
        N = voigt.shape[0]
        out = torch.empty((3 if strain else 6, N), dtype=torch.float64, device=voigt.device)
        _lib.check(lib.fs_stress_invariants(_lib.ptr(voigt), N, int(self._solid), int(strain), _lib.ptr(out),
                                            _lib.stream_ptr()), "fs_stress_invariants")
        return out.cpu().numpy()

    # ------------------------------------------------------------------ results
    def _build_stress_result(self, sigma, is_3d) -> StressResult:
        inv = self._invariants(sigma, strain=False)
        s = sigma.cpu().numpy()
        return StressResult(sigma_xx=s[:, 0], sigma_yy=s[:, 1], sigma_xy=s[:, 3], von_mises=inv[0], sigma_1=inv[1],
                            sigma_2=inv[2], tau_max=inv[4], principal_angle=inv[5],
                            sigma_zz=s[:, 2] if is_3d else None, tau_yz=s[:, 4] if is_3d else None,
                            tau_zx=s[:, 5] if is_3d else None, sigma_3=inv[3] if is_3d else None)

    def _build_strain_result(self, eps, is_3d) -> StrainResult:
        inv = self._invariants(eps, strain=True)
        e = eps.cpu().numpy()
        return StrainResult(epsilon_xx=e[:, 0], epsilon_yy=e[:, 1], gamma_xy=e[:, 3], epsilon_1=inv[0],
                            epsilon_2=inv[1], gamma_max=inv[2], epsilon_zz=e[:, 2] if is_3d else None,
                            gamma_yz=e[:, 4] if is_3d else None, gamma_zx=e[:, 5] if is_3d else None, epsilon_3=None)

    @staticmethod
    def _check_smoothing(smoothing):
        return smoothing == "area_weighted"      # any other string averages with weight 1 (:829-832)

    # ------------------------------------------------------------------ public API
    def compute_element_stresses(self, location=StressLocation.MIDDLE, stress_type=StressType.TOTAL,
                                 gauss_point: Tuple[float, float] = (0.0, 0.0)) -> StressResult:
        """stress_recovery.py:710-766."""
        if self._solid:
            pg = self._solid_values(_lib.FS_SOLID_STRESS_MEAN, False, False)
        else:
            pg = self._shell_points([gauss_point], location, _PARTS[stress_type], False, False)
        return self._build_stress_result(self._element_rows(pg), self._solid)

    def compute_nodal_stresses(self, location=StressLocation.MIDDLE, stress_type=StressType.TOTAL,
                               smoothing: str = "average") -> StressResult:
        """stress_recovery.py:771-866."""
        wt = self._check_smoothing(smoothing)
        if self._solid:
            return self._build_stress_result(self._nodal(self._solid_values(_lib.FS_SOLID_STRESS_NODES, False, wt), 6, wt), True)
        pg = self._shell_points(_SHELL_NODE_COORDS, location, _PARTS[stress_type], False, wt)
        return self._build_stress_result(self._nodal(pg, 3, wt), False)

    def compute_nodal_stresses_all_layers(self, stress_type=StressType.TOTAL, smoothing="average"):
        """stress_recovery.py:871-914."""
        return {"TOP": self.compute_nodal_stresses(StressLocation.TOP, stress_type, smoothing),
                "MID": self.compute_nodal_stresses(StressLocation.MIDDLE, stress_type, smoothing),
                "BOT": self.compute_nodal_stresses(StressLocation.BOTTOM, stress_type, smoothing)}

    def compute_nodal_stresses_all_layers_dict(self, stress_type=StressType.TOTAL, smoothing="average"):
        """stress_recovery.py:916-953."""
        out = {}
        for prefix, result in self.compute_nodal_stresses_all_layers(stress_type, smoothing).items():
            for key, arr in result.to_dict().items():
                out[f"{prefix}_{key}"] = arr
        return out

    def compute_element_strains(self, location=StressLocation.MIDDLE,
                                gauss_point: Tuple[float, float] = (0.0, 0.0)) -> StrainResult:
        """stress_recovery.py:958-1006."""
        if self._solid:
            pg = self._solid_values(_lib.FS_SOLID_STRESS_MEAN, True, False)
        else:
            pg = self._shell_points([gauss_point], location, 3, True, False)
        return self._build_strain_result(self._element_rows(pg), self._solid)

    def compute_nodal_strains(self, location=StressLocation.MIDDLE, smoothing: str = "average") -> StrainResult:
        """stress_recovery.py:1008-1081."""
        wt = self._check_smoothing(smoothing)
        if self._solid:
            return self._build_strain_result(self._nodal(self._solid_values(_lib.FS_SOLID_STRESS_NODES, True, wt), 6, wt), True)
        pg = self._shell_points(_SHELL_NODE_COORDS, location, 3, True, wt)
        return self._build_strain_result(self._nodal(pg, 3, wt), False)


# ---------------------------------------------------------------------- module-level helpers
def compute_von_mises(sigma_xx, sigma_yy, sigma_xy, sigma_zz=None, tau_yz=None, tau_zx=None):
    """stress_recovery.py:1405-1458 (array helper on caller-owned NumPy data, as in the reference)."""
    if sigma_zz is None:
        return np.sqrt(sigma_xx ** 2 + sigma_yy ** 2 - sigma_xx * sigma_yy + 3 * sigma_xy ** 2)
    szz = np.asarray(sigma_zz)
    tyz = np.asarray(tau_yz) if tau_yz is not None else np.zeros_like(szz)
    tzx = np.asarray(tau_zx) if tau_zx is not None else np.zeros_like(szz)
    return np.sqrt(0.5 * ((sigma_xx - sigma_yy) ** 2 + (sigma_yy - szz) ** 2 + (szz - sigma_xx) ** 2)
                   + 3.0 * (sigma_xy ** 2 + tyz ** 2 + tzx ** 2))


def compute_principal_stresses(sigma_xx, sigma_yy, sigma_xy, sigma_zz=None, tau_yz=None, tau_zx=None):
    """stress_recovery.py:1461-1506."""
    if sigma_zz is None:
        avg, diff = (sigma_xx + sigma_yy) / 2, (sigma_xx - sigma_yy) / 2
        R = np.sqrt(diff ** 2 + sigma_xy ** 2)
        return avg + R, avg - R, 0.5 * np.arctan2(2 * sigma_xy, sigma_xx - sigma_yy)
    n = len(sigma_xx)
    s = np.column_stack([sigma_xx, sigma_yy, np.asarray(sigma_zz), sigma_xy,
                         np.asarray(tau_yz) if tau_yz is not None else np.zeros(n),
                         np.asarray(tau_zx) if tau_zx is not None else np.zeros(n)])
    lib = _lib.load()
    from .batch import _dev
    sd = torch.from_numpy(np.ascontiguousarray(s, dtype=np.float64)).to(_dev())
    out = torch.empty((6, n), dtype=torch.float64, device=sd.device)
    _lib.check(lib.fs_stress_invariants(_lib.ptr(sd), n, 1, 0, _lib.ptr(out), _lib.stream_ptr()), "fs_stress_invariants")
    o = out.cpu().numpy()
    return o[1], o[2], o[3]


def compute_max_shear_stress(sigma_xx, sigma_yy, sigma_xy, sigma_zz=None, tau_yz=None, tau_zx=None):
    """stress_recovery.py:1509-1542."""
    if sigma_zz is None:
        return np.sqrt(((sigma_xx - sigma_yy) / 2) ** 2 + sigma_xy ** 2)
    s1, _, s3 = compute_principal_stresses(sigma_xx, sigma_yy, sigma_xy, sigma_zz, tau_yz, tau_zx)
    return (s1 - s3) / 2
