"""Material records consumed by the hot path (mirror of core/material.py:5-141 of the reference:
same class names, field names and factory behaviour; boundary types only, no arithmetic)."""
from dataclasses import dataclass
from typing import Optional, Tuple, Union


@dataclass
class IsotropicMaterial:
    name: str
    E: float
    nu: float
    rho: float
    shear_correction_factor: Optional[float] = None


@dataclass
class OrthotropicMaterial:
    name: str
    E: Tuple[float, float, float]      # (E1, E2, E3)
    G: Tuple[float, float, float]      # (G12, G23, G31)
    nu: Tuple[float, float, float]     # (nu12, nu23, nu31)
    rho: float
    shear_correction_factor: Optional[float] = None

    E1 = property(lambda self: self.E[0])
    E2 = property(lambda self: self.E[1])
    E3 = property(lambda self: self.E[2])
    G12 = property(lambda self: self.G[0])
    G23 = property(lambda self: self.G[1])
    G31 = property(lambda self: self.G[2])
    nu12 = property(lambda self: self.nu[0])
    nu23 = property(lambda self: self.nu[1])
    nu31 = property(lambda self: self.nu[2])


MaterialType = Union[IsotropicMaterial, OrthotropicMaterial]


def _triple(value, label):
    try:
        items = tuple(value)
    except TypeError as exc:
        raise ValueError(f"{label} must be an iterable with 3 entries") from exc
    if len(items) != 3:
        raise ValueError(f"{label} must have exactly 3 components")
    return items


def Material(*, name: str = "Material", E, nu, rho: float, G=None) -> MaterialType:
    """Scalar E/nu -> IsotropicMaterial; length-3 E/nu(+G) -> OrthotropicMaterial
    (core/material.py:102-141)."""
    def vec(v):
        return hasattr(v, "__iter__") and not isinstance(v, (str, bytes))
    if vec(E) or vec(nu):
        Ev, nuv = _triple(E, "E"), _triple(nu, "nu")
        if G is None:
            raise ValueError("Orthotropic materials require shear moduli G")
        return OrthotropicMaterial(name=name, E=Ev, G=_triple(G, "G"), nu=nuv, rho=rho)
    return IsotropicMaterial(name=name, E=float(E), nu=float(nu), rho=float(rho))
