"""Shared test helpers: seeded element generators and property tables."""
import numpy as np

import fem_shell_amd as fsa

BASE_QUAD = np.array([[0, 0, 0], [1, 0, .05], [1.1, 1, 0], [0, 1, .1]], float)

T300 = dict(E=(181e9, 10.3e9, 10.3e9), G=(7.17e9, 3.0e9, 7.17e9), nu=(0.28, 0.28, 0.28), rho=1600.0)


def t300():
    return fsa.OrthotropicMaterial("T300/5208", T300["E"], T300["G"], T300["nu"], T300["rho"])


def standard_props():
    """A mixed property table: isotropic (Ko-2017 materials), symmetric and asymmetric laminates."""
    m = t300()
    lam = lambda ang, t=0.125e-3: fsa.create_laminate_from_angles(m, t, ang)
    return [
        fsa.ShellProperty(fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0), 0.01),
        fsa.ShellProperty(fsa.IsotropicMaterial("scordelis", 4.32e8, 0.0, 360.0), 0.25),
        fsa.ShellProperty(fsa.IsotropicMaterial("hemi", 6.825e7, 0.3, 1.0), 0.04),
        fsa.CompositeShellProperty(lam([0, 90, 90, 0])),
        fsa.CompositeShellProperty(lam([0, 90])),
        fsa.CompositeShellProperty(lam([0, 45, 90])),
        fsa.CompositeShellProperty(lam([30], 1e-3)),
        fsa.CompositeShellProperty(lam([0, 45, -45, 90, 90, -45, 45, 0], 2.5e-3)),
    ]


def random_quads(n, seed=0, far=True):
    """Warped / sheared / scaled / translated quads (the L/h-stressing cases of SURVEY.md A.1)."""
    rng = np.random.default_rng(seed)
    xyz = np.empty((n, 4, 3))
    for t in range(n):
        off = rng.choice([0, 0, 1e3, 3.0] if far else [0.0]) * rng.standard_normal(3)
        scale = rng.choice([1, 1e-3, 20])
        q = BASE_QUAD + 0.1 * rng.standard_normal((4, 3))
        if t % 7 == 0:
            q = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], float)       # unit square
        if t % 7 == 1:
            q = np.array([[0, 0, 0], [2, 0, 0], [2.5, 1, 0], [.5, 1, 0]], float)    # parallelogram
        xyz[t] = q * scale + off
    return xyz
