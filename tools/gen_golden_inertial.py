"""Golden fixture for the inertial nodal forces: runs the UNMODIFIED InertialForcesCalculator of the reference
(solvers/fsi/corotational.py:304-562) on a seeded node cloud -> tests/golden/inertial_ref.npz.  Build container only
(needs /root/reference); the module is loaded by file path because its package __init__ pulls petsc4py, it needs
NumPy alone.   usage: python tools/gen_golden_inertial.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def gen_inertial():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_corotational", ref_shim.REF + "/solvers/fsi/corotational.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(21)
    n = 257
    coords = rng.uniform(-30.0, 60.0, (n, 3))
    vel = rng.standard_normal((n, 3))
    mass = rng.uniform(0.1, 50.0, n)
    axis, center, omega, alpha = np.array([0.3, -0.2, 1.0]), np.array([1.0, 2.0, -0.5]), 0.79, 0.05
    calc = mod.InertialForcesCalculator(axis, center)
    F, diag = calc.compute_all_inertial_forces(coords, vel, mass, omega, alpha)
    out = dict(coords=coords, vel=vel, mass=mass, axis=axis, center=center, omega=omega, alpha=alpha,
               F_cf=calc.compute_centrifugal_force(coords, mass, omega), F_cor=calc.compute_coriolis_force(vel, mass, omega),
               F_eul=calc.compute_euler_force(coords, mass, alpha), F_all=F)
    for k, v in diag.items():
        out["diag_" + k] = np.array([v["total"], v["max_nodal"], v["mean_nodal"]])
    np.savez_compressed(os.path.join(GOLD, "inertial_ref.npz"), **out)
    print("inertial_ref.npz", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    gen_inertial()
