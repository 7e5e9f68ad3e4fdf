"""Classical-lamination-theory property rows (host side, once per distinct layup).

Restates core/laminate.py of the reference (Ply :67-108, compute_Q :111-150, compute_Qbar
:153-204, compute_shear_Cbar :207-241, Laminate :244-534, builders :537-607) with the same public
names.  The per-ply accumulation order is kept (bottom ply first) so A, B, D and Cs are the same
floating-point numbers the reference produces; they become rows of the device property table.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .material import OrthotropicMaterial


@dataclass
class StrengthProperties:
    Xt: float
    Xc: float
    Yt: float
    Yc: float
    S12: float
    S23: Optional[float] = None

    def __post_init__(self):
        if self.S23 is None:
            self.S23 = self.S12 / 2


@dataclass
class Ply:
    material: OrthotropicMaterial
    thickness: float
    angle: float  # degrees
    strength: Optional[StrengthProperties] = None
    z_bottom: float = field(default=0.0, init=False)
    z_top: float = field(default=0.0, init=False)

    @property
    def angle_rad(self) -> float:
        return np.radians(self.angle)


def compute_Q(material: OrthotropicMaterial) -> np.ndarray:
    """Reduced stiffness in material axes (core/laminate.py:136-150)."""
    E1, E2, _ = material.E
    nu12 = material.nu[0]
    G12 = material.G[0]
    nu21 = nu12 * E2 / E1
    denom = 1 - nu12 * nu21
    q12 = nu12 * E2 / denom
    return np.array([[E1 / denom, q12, 0], [q12, E2 / denom, 0], [0, 0, G12]])


def compute_Qbar(material: OrthotropicMaterial, theta_deg: float) -> np.ndarray:
    """Reduced stiffness rotated by the fibre angle (core/laminate.py:182-204)."""
    Q = compute_Q(material)
    Q11, Q12, Q22, Q66 = Q[0, 0], Q[0, 1], Q[1, 1], Q[2, 2]
    theta = np.radians(theta_deg)
    c = np.cos(theta)
    s = np.sin(theta)
    c2, s2 = c**2, s**2
    c3, s3 = c**3, s**3
    c4, s4 = c**4, s**4
    out = np.zeros((3, 3))
    out[0, 0] = Q11 * c4 + 2 * (Q12 + 2 * Q66) * s2 * c2 + Q22 * s4
    out[1, 1] = Q11 * s4 + 2 * (Q12 + 2 * Q66) * s2 * c2 + Q22 * c4
    out[0, 1] = out[1, 0] = (Q11 + Q22 - 4 * Q66) * s2 * c2 + Q12 * (s4 + c4)
    out[0, 2] = out[2, 0] = (Q11 - Q12 - 2 * Q66) * s * c3 + (Q12 - Q22 + 2 * Q66) * s3 * c
    out[1, 2] = out[2, 1] = (Q11 - Q12 - 2 * Q66) * s3 * c + (Q12 - Q22 + 2 * Q66) * s * c3
    out[2, 2] = (Q11 + Q22 - 2 * Q12 - 2 * Q66) * s2 * c2 + Q66 * (s4 + c4)
    return out


def compute_shear_Cbar(material: OrthotropicMaterial, theta_deg: float) -> np.ndarray:
    """[[C55, C45], [C45, C44]] rotated transverse shear moduli (core/laminate.py:230-241)."""
    _, G23, G13 = material.G
    theta = np.radians(theta_deg)
    c = np.cos(theta)
    s = np.sin(theta)
    c2, s2 = c**2, s**2
    C44 = G23 * c2 + G13 * s2
    C55 = G13 * c2 + G23 * s2
    C45 = (G13 - G23) * s * c
    return np.array([[C55, C45], [C45, C44]])


@dataclass
class Laminate:
    plies: List[Ply]
    shear_correction_factor: float = 0.75
    A: np.ndarray = field(init=False, repr=False)
    B: np.ndarray = field(init=False, repr=False)
    D: np.ndarray = field(init=False, repr=False)
    Cs: np.ndarray = field(init=False, repr=False)

    def __post_init__(self):
        if len(self.plies) == 0:
            raise ValueError("Laminate must have at least one ply")
        self._stack()
        self._abd()
        self._shear()

    @property
    def total_thickness(self) -> float:
        return sum(p.thickness for p in self.plies)

    @property
    def n_plies(self) -> int:
        return len(self.plies)

    def _stack(self):  # core/laminate.py:326-334
        z = -self.total_thickness / 2
        for p in self.plies:
            p.z_bottom = z
            z += p.thickness
            p.z_top = z

    def _abd(self):  # core/laminate.py:336-356
        self.A = np.zeros((3, 3))
        self.B = np.zeros((3, 3))
        self.D = np.zeros((3, 3))
        for p in self.plies:
            Qb = compute_Qbar(p.material, p.angle)
            zb, zt = p.z_bottom, p.z_top
            self.A += Qb * (zt - zb)
            self.B += 0.5 * Qb * (zt**2 - zb**2)
            self.D += (1 / 3) * Qb * (zt**3 - zb**3)

    def _shear(self):  # core/laminate.py:358-452
        k = self.shear_correction_factor
        A44 = A45 = A55 = 0.0
        for p in self.plies:
            Cb = compute_shear_Cbar(p.material, p.angle)
            A55 += Cb[0, 0] * p.thickness
            A45 += Cb[0, 1] * p.thickness
            A44 += Cb[1, 1] * p.thickness
        if len(self.plies) == 1:
            self.Cs = k * np.array([[A55, A45], [A45, A44]])
            return
        D11, D22 = self.D[0, 0], self.D[1, 1]
        flex = [0.0, 0.0]   # direction 55 (uses Q11), direction 44 (uses Q22)
        cum = [0.0, 0.0]
        for p in self.plies:
            Qb = compute_Qbar(p.material, p.angle)
            Cb = compute_shear_Cbar(p.material, p.angle)
            zb, zt, t = p.z_bottom, p.z_top, p.thickness
            dz3 = (zt**3 - zb**3) / 3
            dz5 = (zt**5 - zb**5) / 5
            for d, (Qaa, Caa) in enumerate(((Qb[0, 0], Cb[0, 0]), (Qb[1, 1], Cb[1, 1]))):
                a0 = cum[d] - Qaa / 2 * zb**2     # Phi(z) = a0 + b0 z^2 inside the ply
                b0 = Qaa / 2
                integ = a0**2 * t + 2 * a0 * b0 * dz3 + b0**2 * dz5
                if Caa > 0:
                    flex[d] += integ / Caa
                cum[d] += Qaa * (zt**2 - zb**2) / 2
        Cs55 = D11**2 / flex[0] if (flex[0] > 0 and D11 > 0) else k * A55
        Cs44 = D22**2 / flex[1] if (flex[1] > 0 and D22 > 0) else k * A44
        if A55 > 0 and A44 > 0:
            Cs45 = np.sqrt((Cs55 / A55) * (Cs44 / A44)) * A45
        else:
            Cs45 = k * A45
        self.Cs = np.array([[Cs55, Cs45], [Cs45, Cs44]])

    @property
    def is_symmetric(self) -> bool:  # core/laminate.py:454-461
        return bool(np.allclose(self.B, 0, atol=1e-10 * np.max(np.abs(self.A))))

    @property
    def is_balanced(self) -> bool:  # core/laminate.py:463-471
        tol = 1e-10 * np.max(np.abs(self.A))
        return bool(np.abs(self.A[0, 2]) < tol and np.abs(self.A[1, 2]) < tol)

    def get_ABD_matrix(self) -> np.ndarray:
        return np.block([[self.A, self.B], [self.B, self.D]])

    def get_equivalent_properties(self) -> dict:  # core/laminate.py:489-530
        h = self.total_thickness
        a = np.linalg.inv(self.A)
        d = np.linalg.inv(self.D)
        return {
            "Ex_membrane": 1 / (h * a[0, 0]), "Ey_membrane": 1 / (h * a[1, 1]),
            "Gxy_membrane": 1 / (h * a[2, 2]), "nuxy_membrane": -a[0, 1] / a[0, 0],
            "Ex_bending": 12 / (h**3 * d[0, 0]), "Ey_bending": 12 / (h**3 * d[1, 1]),
            "total_thickness": h,
        }

    def __repr__(self) -> str:
        angles = "/".join(f"{p.angle:.0f}" for p in self.plies)
        return f"Laminate([{angles}], h={self.total_thickness * 1000:.3f}mm)"


def create_symmetric_laminate(half_plies: List[Ply], shear_correction_factor: float = 0.75) -> Laminate:
    import copy
    return Laminate(half_plies + [copy.deepcopy(p) for p in reversed(half_plies)],
                    shear_correction_factor)


def create_laminate_from_angles(material, ply_thickness, angles, strength=None,
                                shear_correction_factor: float = 0.75) -> Laminate:
    return Laminate([Ply(material, ply_thickness, a, strength) for a in angles],
                    shear_correction_factor)


def quasi_isotropic_layup(n_repeats: int = 1) -> List[float]:
    half = [0, 45, -45, 90] * n_repeats
    return half + half[::-1]


def cross_ply_layup(n_plies: int = 4) -> List[float]:
    if n_plies % 4 != 0:
        raise ValueError("n_plies must be multiple of 4 for symmetric cross-ply")
    half = [0 if i % 2 == 0 else 90 for i in range(n_plies // 2)]
    return half + half[::-1]


def angle_ply_layup(angle: float, n_plies: int = 4) -> List[float]:
    if n_plies % 4 != 0:
        raise ValueError("n_plies must be multiple of 4 for symmetric angle-ply")
    half = [angle if i % 2 == 0 else -angle for i in range(n_plies // 2)]
    return half + half[::-1]
