"""TEST INFRASTRUCTURE ONLY (oracle): NumPy restatement of the reference's InertialForcesCalculator
(solvers/fsi/corotational.py:347-552) -- centrifugal m w^2 r_perp, Coriolis -2 m (w x v), Euler -m (alpha x r) -- pinned
by tests/golden/inertial_ref.npz (generated from the unmodified reference, tools/gen_golden_inertial.py)."""
import numpy as np


def inertial_forces(coords, vel, mass, omega, alpha, axis, center, centrifugal=True, coriolis=True, euler=True):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)                       # corotational.py:336-340
    r = np.asarray(coords, dtype=np.float64) - np.asarray(center, dtype=np.float64)
    m = np.asarray(mass, dtype=np.float64)[:, None]
    F = np.zeros_like(r)
    parts = {}
    if centrifugal and omega != 0.0:                         # :374-388
        r_perp = r - np.outer(r @ axis, axis)
        parts["centrifugal"] = m * (omega * omega) * r_perp
    if coriolis and omega != 0.0:                            # :418-429
        parts["coriolis"] = -2.0 * m * np.cross(omega * axis, np.asarray(vel, dtype=np.float64))
    if euler and alpha != 0.0:                               # :456-469
        parts["euler"] = -m * np.cross(alpha * axis, r)
    for k in ("centrifugal", "coriolis", "euler"):           # summed in this order (:515-543)
        if k in parts:
            F += parts[k]
    return F, parts
