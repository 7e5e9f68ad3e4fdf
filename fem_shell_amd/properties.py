"""Shell property objects + the device property table.

ShellProperty / CompositeShellProperty mirror core/properties.py:10-62 of the reference.
build_prop_table turns a list of properties into rows of `fs_shell_prop` (include/femshell_hip.h)
using the reference's own NumPy expressions for the constitutive constants so the numbers are
bit-identical: MITC4.py:671-734 (isotropic), MITC4_composite.py:172-204 (laminate).
"""
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np

from .laminate import Laminate
from .material import IsotropicMaterial, MaterialType, OrthotropicMaterial

FS_PROP_COUPLED = 1
FS_PROP_LAMINATE = 2

PROP_DTYPE = np.dtype([
    ("A", "f8", (9,)), ("B", "f8", (9,)), ("D", "f8", (9,)), ("Cs", "f8", (4,)),
    ("h", "f8"), ("m_t", "f8"), ("m_r", "f8"), ("k_drill", "f8"), ("C_raw", "f8", (9,)),
    ("rho", "f8"), ("flags", "i4"), ("nonlinear", "i4"),
])
assert PROP_DTYPE.itemsize == 368


@dataclass
class ShellProperty:
    material: MaterialType
    thickness: float

    @property
    def total_thickness(self) -> float:
        return self.thickness

    def to_element_kwargs(self) -> dict:
        return {"material": self.material, "thickness": self.thickness}


@dataclass
class CompositeShellProperty:
    laminate: Laminate

    @property
    def total_thickness(self) -> float:
        return self.laminate.total_thickness

    def to_element_kwargs(self) -> dict:
        return {"laminate": self.laminate}


ShellPropertyType = Union[ShellProperty, CompositeShellProperty]


def iso_row(material, thickness, shear_correction_factor: Optional[float] = None,
            nonlinear: bool = False) -> np.ndarray:
    """One fs_shell_prop row for an isotropic shell (MITC4.py:77-85, 671-734)."""
    if isinstance(material, OrthotropicMaterial):
        # MITC4.Cm raises for orthotropic single-layer shells (MITC4.py:673-674)
        raise NotImplementedError("Orthotropic material not yet supported in MITC4")
    row = np.zeros((), dtype=PROP_DTYPE)
    E, nu, h = material.E, material.nu, thickness
    if shear_correction_factor is not None:
        k = shear_correction_factor
    elif getattr(material, "shear_correction_factor", None) is not None:
        k = material.shear_correction_factor
    else:
        k = 5.0 / 6.0
    shape = np.array([[1, nu, 0], [nu, 1, 0], [0, 0, (1 - nu) / 2]])
    row["A"] = (E * h / (1 - nu**2) * shape).ravel()
    row["D"] = (E * h**3 / (12 * (1 - nu**2)) * shape).ravel()
    G = E / (2 * (1 + nu))
    row["Cs"] = (k * G * h * np.eye(2)).ravel()
    row["h"] = h
    row["m_t"] = material.rho * h
    row["m_r"] = material.rho * h**3 / 12.0
    row["k_drill"] = E * (h**2) * 0.15
    row["C_raw"] = (E / (1 - nu**2) * shape).ravel()
    row["rho"] = material.rho
    row["flags"] = 0
    row["nonlinear"] = int(bool(nonlinear))
    return row


def laminate_row(lam: Laminate, nonlinear: bool = False) -> np.ndarray:
    """One fs_shell_prop row for a laminate (MITC4_composite.py:90-117, 172-208)."""
    row = np.zeros((), dtype=PROP_DTYPE)
    h = lam.total_thickness
    row["A"] = lam.A.ravel()
    row["B"] = lam.B.ravel()
    row["D"] = lam.D.ravel()
    row["Cs"] = lam.Cs.ravel()
    row["h"] = h
    row["m_t"] = sum(p.material.rho * p.thickness for p in lam.plies)
    row["m_r"] = sum(p.material.rho * (p.z_top**3 - p.z_bottom**3) / 3 for p in lam.plies)
    row["k_drill"] = np.trace(lam.A) / 3.0 * h * 0.15
    row["C_raw"] = (lam.A / h).ravel()
    row["rho"] = lam.plies[0].material.rho   # equivalent material of the composite classes (MITC3_composite.py:121,133)
    row["flags"] = FS_PROP_LAMINATE | (0 if lam.is_symmetric else FS_PROP_COUPLED)
    row["nonlinear"] = int(bool(nonlinear))
    return row


def prop_row(prop, nonlinear: bool = False, shear_correction_factor=None) -> np.ndarray:
    if isinstance(prop, CompositeShellProperty):
        return laminate_row(prop.laminate, nonlinear)
    if isinstance(prop, Laminate):
        return laminate_row(prop, nonlinear)
    if isinstance(prop, ShellProperty):
        return iso_row(prop.material, prop.thickness, shear_correction_factor, nonlinear)
    raise TypeError(f"unsupported shell property {type(prop).__name__}")


def build_prop_table(props: Sequence, nonlinear: bool = False,
                     shear_correction_factor=None) -> np.ndarray:
    """Structured array [P] of fs_shell_prop rows (host memory; copied to the device once)."""
    out = np.zeros(len(props), dtype=PROP_DTYPE)
    for i, p in enumerate(props):
        out[i] = prop_row(p, nonlinear, shear_correction_factor)
    return out


# ---------------------------------------------------------------------------------------------
# solids: rows of [C (36, global axes), rho]  (fs_solid_batch)
# ---------------------------------------------------------------------------------------------
_VOIGT = ((0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (2, 0))     # xx, yy, zz, xy, yz, zx


def solid_C(material, orientation=None) -> np.ndarray:
    """6x6 constitutive matrix of a solid element in global axes (elements/SOLID.py:377-541):
    isotropic Lame form, or orthotropic inv(S) rotated as T C T^T with the Voigt stress
    transformation T of the 3x3 `orientation`."""
    if isinstance(material, IsotropicMaterial):
        E, nu = material.E, material.nu
        lambd = E * nu / ((1 + nu) * (1 - 2 * nu))
        mu = E / (2 * (1 + nu))
        C = np.zeros((6, 6))
        C[:3, :3] = lambd
        C[np.arange(3), np.arange(3)] = lambd + 2 * mu
        C[np.arange(3, 6), np.arange(3, 6)] = mu
        return C
    if not isinstance(material, OrthotropicMaterial):
        raise TypeError(f"Unsupported material type: {type(material)}")
    E1, E2, E3 = material.E
    G12, G23, G31 = material.G
    nu12, nu23, nu31 = material.nu
    nu21, nu32, nu13 = nu12 * E2 / E1, nu23 * E3 / E2, nu31 * E1 / E3
    S = np.zeros((6, 6))
    S[0, :3] = (1 / E1, -nu21 / E2, -nu31 / E3)
    S[1, :3] = (-nu12 / E1, 1 / E2, -nu32 / E3)
    S[2, :3] = (-nu13 / E1, -nu23 / E2, 1 / E3)
    S[3, 3], S[4, 4], S[5, 5] = 1 / G12, 1 / G23, 1 / G31
    C = np.linalg.inv(S)
    if orientation is None:
        return C
    r = np.asarray(orientation, dtype=float)
    T = np.empty((6, 6))
    for I, (i, j) in enumerate(_VOIGT):
        for J, (k, l) in enumerate(_VOIGT):
            if J < 3:
                T[I, J] = r[i, k] ** 2 if I < 3 else r[i, k] * r[j, k]
            else:
                T[I, J] = 2 * r[i, k] * r[i, l] if I < 3 else r[i, k] * r[j, l] + r[i, l] * r[j, k]
    return T @ C @ T.T


def solid_row(material, orientation=None) -> np.ndarray:
    return np.concatenate([solid_C(material, orientation).ravel(), [float(material.rho)]])
